"""Stand-alone timing of the GruBlock forward at B = 64: fused projection + recurrence vs GEMM + recurrence.  python tools/micro_gru_fused.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tatt_amd.lib import lib  # noqa: E402  (needs a library built with tools/experiments/gru_fused.hip added to the Makefile)

L = lib()
B, H, W = 64, 16, 64
Rr = B * H * W
VP = ctypes.c_void_p


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


for vertical, nin in ((True, 2), (False, 1)):
    Cin = 64 * nin
    xs = [torch.randn(Rr, 64, device="cuda") for _ in range(nin)]
    Wf, bf = torch.randn(192, Cin, device="cuda") * 0.15, torch.randn(192, device="cuda") * 0.1
    whh = [torch.randn(96, 32, device="cuda") * 0.3 for _ in range(2)]
    bhh = [torch.randn(96, device="cuda") * 0.1 for _ in range(2)]
    sm = (B * W, H, W, H * W, 1, W) if vertical else (B * H, W, 1, W, 0, 1)
    gi = torch.empty(Rr, 192, device="cuda")
    out, sv = torch.empty(Rr, 64, device="cuda"), torch.empty(Rr, 2, 4, 32, device="cuda")
    Wp = torch.empty(192 * Cin, device="cuda")
    L.gru_pack_wf(1, (VP * 1)(Wf.data_ptr()), (VP * 1)(Wp.data_ptr()), (ctypes.c_int * 1)(Cin))
    x2 = xs[1] if nin > 1 else None
    t_g = timeit(lambda: L.gemm_cat(xs[0], x2, 64, Wf, gi, None, 0, bf, Rr, 192, Cin, 64, Cin, 1, 192, 0, 0, 0, 0))
    t_r = timeit(lambda: L.gru32_fwd(gi, whh[0], whh[1], bhh[0], bhh[1], out, sv, *sm))
    t_f = timeit(lambda: L.gru32_proj_fwd(xs[0], x2, Wp, bf, whh[0], whh[1], bhh[0], bhh[1], out, sv, *sm))
    t_fn = timeit(lambda: L.gru32_proj_fwd(xs[0], x2, Wp, bf, whh[0], whh[1], bhh[0], bhh[1], out, None, *sm))
    print(f"vertical={vertical} Cin={Cin}: gemm_cat {t_g:.1f} us + gru32_fwd {t_r:.1f} us = {t_g + t_r:.1f};  fused {t_f:.1f} us  (without save {t_fn:.1f})")
