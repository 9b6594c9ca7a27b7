"""Does tatt_dec_bwd give bitwise the same result when another kernel runs beside it on a second stream?
python tools/probe_corun.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tatt_amd.lib import lib  # noqa: E402

L = lib()
B, Lq = 64, 1024
g = torch.Generator().manual_seed(1)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).cuda()
R0, Rt = B * Lq, B * 26
VP, U32 = ctypes.c_void_p, ctypes.c_uint32
new = lambda *s: torch.empty(*s, device="cuda")
tgt, qp = rn(R0, 64), rn(R0, 64, sc=0.5)
nw, nb = 1 + rn(64, sc=0.1), rn(64, sc=0.1)
d = []
for _ in range(2):
    p = {"Kp": rn(Rt, 64), "Vp": rn(Rt, 64)}
    for w, b_ in (("Wq", "bq"), ("Wo", "bo"), ("W1", "b1"), ("W2", "b2")):
        p[w], p[b_] = rn(64, 64, sc=0.15), rn(64, sc=0.1)
    for w, b_ in (("g2", "be2"), ("g3", "be3")):
        p[w], p[b_] = 1 + rn(64, sc=0.1), rn(64, sc=0.1)
    d.append(p)
packed = new(L.dec_packed_w_floats())
L.dec_pack_w((VP * 8)(*[p[k].data_ptr() for p in d for k in ("Wq", "Wo", "W1", "W2")]), packed)
tabs = []
for p in d:
    t = new(L.dec_kv_table_floats(B))
    L.dec_pack_kv(p["Kp"], p["Vp"], t, B)
    tabs.append(t)
params = [[p[k] for k in ("bq", "bo", "b1", "b2", "g2", "be2", "g3", "be3")] for p in d]
saved = [[new(R0, 64) for _ in range(7)] for _ in range(2)]
tp = new(R0, 64)
sites = (U32 * 8)(*range(8))
L.dec_fwd(tgt, qp, packed, (VP * 2)(*[t.data_ptr() for t in tabs]), (VP * 16)(*[t.data_ptr() for ps in params for t in ps]), nw, nb,
          (VP * 14)(*[t.data_ptr() for sv in saved for t in sv]), tp, B, Lq, 1, sites, 0, 1.0, None)
dtp = rn(R0, 64)


def bwd():
    li = 1
    dtgt1, dqp = new(R0, 64), new(R0, 64)
    dY = [new(R0, 64) for _ in range(4)]
    kvpart, lnpart = new(L.dec_bwd_kvpart_floats(B, Lq)), new(L.dec_bwd_lnpart_floats(B, Lq))
    L.dec_bwd(li, None, dtp, packed, tabs[li], (VP * 8)(*[t.data_ptr() for t in params[li]]), nw,
              (VP * 7)(*[t.data_ptr() for t in saved[li]]), dtgt1, 0, dqp, 0,
              (VP * 4)(*[t.data_ptr() for t in dY]), kvpart, lnpart, B, Lq, 1, (U32 * 4)(*range(4, 8)), 0, 1.0, None)
    return [dtgt1, dqp, kvpart, lnpart] + dY


torch.cuda.synchronize()
ref = bwd()
torch.cuda.synchronize()
again = bwd()
torch.cuda.synchronize()
print("solo repeat equal:", all(torch.equal(a, b) for a, b in zip(ref, again)))

# co-runners
X, dYc = rn(R0, 64), rn(R0, 64, sc=0.01)
dW = torch.zeros(64, 64, 3, 3, device="cuda")
ws2 = new(48 * 1024 * 1024)
Wp = rn(64 * 576)
Yc = new(R0, 64)
side = torch.cuda.Stream()


def corun(name, fn, n=6):
    torch.cuda.synchronize()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(n):
            fn()
    outs = []
    for _ in range(3):
        outs.append(bwd())
    torch.cuda.synchronize()
    bad = [sum(0 if torch.equal(a, b) else 1 for a, b in zip(ref, o)) for o in outs]
    mx = max(float((a - b).abs().max()) for o in outs for a, b in zip(ref, o))
    print(f"{name:34s} tensors differing per run {bad}  max|diff| {mx:.3e}")




def fwd_v():
    sv = [[new(R0, 64) for _ in range(7)] for _ in range(2)]
    tp2 = new(R0, 64)
    L.dec_fwd(tgt, qp, packed, (VP * 2)(*[t.data_ptr() for t in tabs]), (VP * 16)(*[t.data_ptr() for ps in params for t in ps]), nw, nb,
              (VP * 14)(*[t.data_ptr() for s_ in sv for t in s_]), tp2, B, Lq, 1, sites, 0, 1.0, None)
    return [tp2] + sv[0] + sv[1]


def conv_v(flag):
    def f():
        y = new(R0, 64)
        L.conv_fwd(X, Wp, None, y, B, 16, 64, 64, 64, 3, 3, 64, 64, 576, 1, 0, 64, 0, flag)
        return [y]
    return f


X2 = rn(R0, 128)
W2 = rn(192, 128)


def gemm_v():
    o = new(R0, 192)
    L.gemm(X2, W2, o, None, R0, 192, 128, 128, 1, 128, 1, 192, 1, 0, 0, 0, 0, 0, 1, None)
    return [o]


def copy_v():
    o = new(R0, 128)
    o.copy_(X2)
    return [o]


def agg():
    L.conv_wgrad(X, dYc, dW, B, 16, 64, 64, 64, 3, 3, 64, 64, 16, ws2, 4)





# ---- other kernels with packed-fp32 VALU code beside the same aggressor
P_ = R0
gi = rn(P_, 192, sc=0.5)
whh = [rn(96, 32, sc=0.2) for _ in range(2)]
bhh = [rn(96, sc=0.1) for _ in range(2)]
H_, W_ = 16, 64
seq = {"v": (B * W_, H_, W_, H_ * W_, 1, W_), "h": (B * H_, W_, 1, W_, 0, 1)}


def gru_f(o):
    def f():
        out, sv = new(P_, 64), new(P_, 2, 4, 32)
        L.gru32_fwd(gi, whh[0], whh[1], bhh[0], bhh[1], out, sv, *seq[o])
        return [out, sv]
    return f


go, gs = gru_f("v")()
go_h, gs_h = gru_f("h")()
dout = rn(P_, 64)


def gru_b(o, out_, sv_):
    def f():
        dgi, dghn, hp = new(P_, 192), new(P_, 64), new(P_, 64)
        L.gru32_bwd(dout, out_, sv_, whh[0], whh[1], dgi, dghn, hp, *seq[o])
        return [dgi, dghn, hp]
    return f


up = rn(4 * R0, 64)
W9 = rn(4, 64, 9, 9, sc=0.05)
wsz = L.conv_skinny_workspace(B, 32, 128, 64, 4, 9, 9)
zbuf = new(wsz)


def c4_v():
    y = new(4 * R0, 4)
    L.conv_skinny_fwd(up, W9, None, y, B, 32, 128, 64, 4, 9, 9, 64, 4, 64 * 81, 81, 0, 1, 0, zbuf)
    return [y]


stat = torch.stack([rn(64, sc=0.1), 1 + rn(64, sc=0.1).abs()]).contiguous()
gam, bet = 1 + rn(64, sc=0.1), rn(64, sc=0.1)


def bn_v():
    y = new(R0, 64)
    L.bn_apply(X, 64, stat, gam, bet, dYc, 64, y, 64, R0, 64, 2)
    return [y]


def ln_v():
    v, y = new(R0, 64), new(R0, 64)
    L.ln_fwd(X, dYc, 0, 0, 1.0, None, gam, bet, v, y, R0, 1e-5)
    return [v, y]


def ps_v():
    y = new(4 * R0, 64)
    L.pixshuf_mish_fwd(up.view(R0, 256), y, B, 16, 64, 64)
    return [y]


for name, vict in (("dec_bwd", bwd), ("dec_fwd", fwd_v), ("gru32_fwd vertical", gru_f("v")), ("gru32_fwd horizontal", gru_f("h")),
                   ("gru32_bwd vertical", gru_b("v", go, gs)), ("gru32_bwd horizontal", gru_b("h", go_h, gs_h)), ("conv 9x9 Cout 4 direct", c4_v),
                   ("bn_apply mish + res", bn_v), ("ln_fwd", ln_v), ("pixshuf_mish_fwd", ps_v)):
    torch.cuda.synchronize()
    r = vict()
    torch.cuda.synchronize()
    tot = []
    for rep in range(4):
        torch.cuda.synchronize()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(8):
                agg()
        outs = [vict() for _ in range(3)]
        torch.cuda.synchronize()
        tot += [sum(0 if torch.equal(a, b) else 1 for a, b in zip(r, o)) for o in outs]
    print(f"victim {name:26s} tensors differing per run {tot} of {len(r)}")
