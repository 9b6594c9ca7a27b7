"""tatt_tn_small_group: several small weight-gradient problems in one launch pair, against fp64 and against tatt_tn_small."""
import ctypes

import pytest
import torch

from tests.util import rel_l2

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("R", [70000, 1664, 37])
def test_group_matches_fp64_and_single(R):
    from tatt_amd.lib import lib
    L = lib()
    big, hp, g_in, extra = rnd(R, 192, seed=1).cuda(), rnd(R, 64, seed=2).cuda(), rnd(R, 64, seed=3).cuda(), rnd(R, 40, seed=4).cuda()
    R2 = max(1, R // 3)                                      # problems of a group may have different row counts
    probs = [  # (dy view, lddy, x view, ldx, M, N, R, with bias)
        (big.view(-1)[96:], 192, hp.view(-1)[32:], 64, 96, 32, R, True),
        (big, 192, g_in, 64, 96, 64, R, True),
        (g_in, 64, hp, 64, 64, 64, R, False),
        (extra, 40, g_in, 64, 40, 64, R2, True),
        (hp, 64, extra, 40, 7, 33, R, True),
    ]
    # second source for the columns >= msplit of problem 0 (the n gate of a GRU direction's W_hh gradient) and of problem 3
    second = {0: (g_in.view(-1)[32:], 64, 64), 3: (hp, 64, 32)}
    ws = torch.empty(L.tn_small_workspace(96, 64), device="cuda")
    dWs = [torch.ones(M, N + 3, device="cuda") for (_, _, _, _, M, N, _, _) in probs]      # row stride lddw > N
    dbs = [torch.ones(M, device="cuda") if wb else None for (*_, M, _, _, wb) in probs]
    n = len(probs)
    VP, LG, IN = ctypes.c_void_p * n, ctypes.c_long * n, ctypes.c_int * n
    L.tn_small_group(n, VP(*[p[0].data_ptr() for p in probs]), LG(*[p[1] for p in probs]), VP(*[p[2].data_ptr() for p in probs]),
                     LG(*[p[3] for p in probs]), VP(*[d.data_ptr() for d in dWs]), LG(*[d.shape[1] for d in dWs]),
                     VP(*[None if b is None else b.data_ptr() for b in dbs]), IN(*[p[4] for p in probs]), IN(*[p[5] for p in probs]),
                     IN(*[p[6] for p in probs]), VP(*[second[i][0].data_ptr() if i in second else None for i in range(n)]),
                     LG(*[second[i][1] if i in second else 0 for i in range(n)]), IN(*[second[i][2] if i in second else 0 for i in range(n)]), ws)
    for i, ((dy, lddy, x, ldx, M, N, Rp, wb), dW, db) in enumerate(zip(probs, dWs, dbs)):
        dyv = torch.as_strided(dy, (Rp, M), (lddy, 1)).cpu().double()
        if i in second:
            d2, ld2, ms = second[i]
            dyv[:, ms:] = torch.as_strided(d2, (Rp, M - ms), (ld2, 1)).cpu().double()
        xv = torch.as_strided(x, (Rp, N), (ldx, 1)).cpu().double()
        assert rel_l2(dW[:, :N].cpu().double().numpy(), (1 + dyv.t() @ xv).numpy()) < 2e-5
        assert torch.equal(dW[:, N:].cpu(), torch.ones(M, 3))
        if wb:
            assert rel_l2(db.cpu().double().numpy(), (1 + dyv.sum(0)).numpy()) < 2e-5
        if i in second:
            continue
        one = torch.ones(M, N + 3, device="cuda")
        L.tn_small(dy, lddy, x, ldx, one, N + 3, None, M, N, Rp, ws)
        assert rel_l2(one[:, :N].cpu().double().numpy(), dW[:, :N].cpu().double().numpy()) < 5e-6


def test_multi_copy_and_grouped_weight_pack():
    """tatt_multi_copy (stacked BiGRU input weights) and tatt_conv_pack_w_group against torch indexing and tatt_conv_pack_w."""
    from tatt_amd.lib import lib
    L = lib()
    srcs = [rnd(96, 64, seed=i).cuda() for i in range(5)] + [rnd(7, seed=9).cuda(), rnd(1, seed=10).cuda()]
    lens = [t.numel() for t in srcs]
    offs = [0]
    for n in lens[:-1]:
        offs.append(offs[-1] + n + 3)                       # gaps stay untouched
    dst = torch.full((offs[-1] + lens[-1] + 5,), -7.0, device="cuda")
    n = len(srcs)
    L.multi_copy(n, (ctypes.c_void_p * n)(*[t.data_ptr() for t in srcs]), (ctypes.c_long * n)(*offs), (ctypes.c_long * n)(*lens), dst)
    want = torch.full_like(dst, -7.0)
    for t, o in zip(srcs, offs):
        want[o:o + t.numel()] = t.reshape(-1)
    assert torch.equal(dst, want)
    shapes = [(64, 64, 3), (256, 64, 3), (32, 4, 3), (64, 4, 9), (5, 7, 1)]
    Ws = [rnd(co, ci, k, k, seed=20 + i).cuda() for i, (co, ci, k) in enumerate(shapes)]
    f1 = [torch.empty(w.numel(), device="cuda") for w in Ws]
    d1 = [torch.empty(w.numel(), device="cuda") for w in Ws]
    f2 = [torch.empty(w.numel(), device="cuda") for w in Ws]
    d2 = [torch.empty(w.numel(), device="cuda") for w in Ws]
    for w, a, b, (co, ci, k) in zip(Ws, f1, d1, shapes):
        L.conv_pack_w(w, a, b, co, ci, k, k)
    n = len(Ws)
    VP, IN = ctypes.c_void_p * n, ctypes.c_int * n
    L.conv_pack_w_group(n, VP(*[w.data_ptr() for w in Ws]), VP(*[t.data_ptr() for t in f2]), VP(*[t.data_ptr() for t in d2]),
                        IN(*[s[0] for s in shapes]), IN(*[s[1] for s in shapes]), IN(*[s[2] * s[2] for s in shapes]))
    for i, (w, (co, ci, k)) in enumerate(zip(Ws, shapes)):
        assert torch.equal(f1[i], f2[i]) and torch.equal(d1[i], d2[i]), i
        assert torch.equal(f2[i].view(co, k * k, ci), w.reshape(co, ci, k * k).permute(0, 2, 1))
        assert torch.equal(d2[i].view(ci, k * k, co), w.reshape(co, ci, k * k).flip(2).permute(1, 2, 0))


def test_conv_tap_unrolled_loop_is_bit_identical_to_generic_loop():
    """The tap-unrolled K loop of the 3x3 Cin=64 convolutions (TAP9) adds the same products in the same order as the generic
    loop: the outputs of forward and input-gradient launches must agree bit for bit (TATT_CONV_TAP9=0 selects the generic loop
    when the library is loaded, hence the two child processes)."""
    import hashlib, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, hashlib, torch
sys.path.insert(0, %r)
from tatt_amd.lib import lib
L = lib(); B, H, W = 5, 16, 64; R = B * H * W
g = torch.Generator().manual_seed(3)
x, w, b = [torch.randn(*s, generator=g).cuda() for s in ((R, 64), (64, 64, 3, 3), (64,))]
w256 = torch.randn(256, 64, 3, 3, generator=g).cuda()
wf, wd = torch.empty(64 * 576, device="cuda"), torch.empty(64 * 576, device="cuda")
L.conv_pack_w(w, wf, wd, 64, 64, 3, 3)
wf2, wd2 = torch.empty(256 * 576, device="cuda"), torch.empty(256 * 576, device="cuda")
L.conv_pack_w(w256, wf2, wd2, 256, 64, 3, 3)
y, dx, y2 = torch.empty(R, 64, device="cuda"), torch.empty(R, 64, device="cuda"), torch.empty(R, 256, device="cuda")
L.conv_fwd(x, wf, b, y, B, H, W, 64, 64, 3, 3, 64, 64, 576, 1, 0, 64, 0, 1)      # packed forward, relu
L.conv_fwd(y, wd, None, dx, B, H, W, 64, 64, 3, 3, 64, 64, 576, 1, 0, 64, 0, 0)  # packed dgrad weights
L.conv_fwd(x, wf2, None, y2, B, H, W, 64, 256, 3, 3, 64, 256, 576, 1, 0, 64, 0, 0)
torch.cuda.synchronize()
h = hashlib.sha256()
for t in (y, dx, y2):
    h.update(t.cpu().numpy().tobytes())
print("HASH", h.hexdigest(), float(y.abs().sum()), float(y2.abs().sum()))
''' % root
    out = {}
    for mode in ("0", "1"):
        env = dict(os.environ, TATT_CONV_TAP9=mode)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("HASH")][0].split()
        assert float(line[2]) > 0 and float(line[3]) > 0
        out[mode] = line[1]
    assert out["0"] == out["1"]
