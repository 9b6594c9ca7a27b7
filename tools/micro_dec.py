"""Stand-alone timing of the fused decoder launches at the benchmark shape (B = 64, Lq = 1024): python tools/micro_dec.py [p_drop]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tatt_amd.lib import lib  # noqa: E402

B, Lq = 64, 1024
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
L = lib()
R0, Rt = B * Lq, B * 26
new = lambda *s: torch.randn(*s, device="cuda")
VP, U32 = ctypes.c_void_p, ctypes.c_uint32
W = [new(64, 64) * 0.15 for _ in range(8)]
packed = torch.empty(L.dec_packed_w_floats(), device="cuda")
L.dec_pack_w((VP * 8)(*[w.data_ptr() for w in W]), packed)
tabs = []
for li in range(2):
    t = torch.empty(L.dec_kv_table_floats(B), device="cuda")
    L.dec_pack_kv(new(Rt, 64), new(Rt, 64), t, B)
    tabs.append(t)
params = [[new(64) * 0.1 for _ in range(4)] + [1 + new(64) * 0.1, new(64) * 0.1, 1 + new(64) * 0.1, new(64) * 0.1] for _ in range(2)]
saved = [[torch.empty(R0, 64, device="cuda") for _ in range(7)] for _ in range(2)]
tgt, qp, tp, nw, nb = new(R0, 64), new(R0, 64) * 0.5, torch.empty(R0, 64, device="cuda"), 1 + new(64) * 0.1, new(64) * 0.1
thresh, scale = int(p * (1 << 24)), 1.0 / (1.0 - p)
sites = (U32 * 8)(*range(1, 9))


def fwd():
    L.dec_fwd(tgt, qp, packed, (VP * 2)(*[t.data_ptr() for t in tabs]), (VP * 16)(*[t.data_ptr() for ps in params for t in ps]), nw, nb,
              (VP * 14)(*[t.data_ptr() for sv in saved for t in sv]), tp, B, Lq, 1, sites, thresh, scale, None)


dtp, dtgt1, dtgt0, dqp = new(R0, 64), torch.empty(R0, 64, device="cuda"), torch.zeros(R0, 64, device="cuda"), torch.empty(R0, 64, device="cuda")
dY = [torch.empty(R0, 64, device="cuda") for _ in range(4)]
kvpart = torch.empty(L.dec_bwd_kvpart_floats(B, Lq), device="cuda")
lnpart = torch.empty(L.dec_bwd_lnpart_floats(B, Lq), device="cuda")


def bwd(li):
    L.dec_bwd(li, None if li == 1 else dtgt1, dtp, packed, tabs[li], (VP * 8)(*[t.data_ptr() for t in params[li]]), nw,
              (VP * 7)(*[t.data_ptr() for t in saved[li]]), dtgt1 if li == 1 else dtgt0, 0 if li == 1 else 1, dqp, 0 if li == 1 else 1,
              (VP * 4)(*[t.data_ptr() for t in dY]), kvpart, lnpart, B, Lq, 1, (U32 * 4)(*range(1 + 4 * li, 5 + 4 * li)), thresh, scale, None)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"p_drop={p}  dec_fwd {timeit(fwd):.1f} us   dec_bwd(1) {timeit(lambda: bwd(1)):.1f} us   dec_bwd(0) {timeit(lambda: bwd(0)):.1f} us")
