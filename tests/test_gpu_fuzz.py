"""Seeded random-shape sweeps of the two workhorse entry points (tatt_gemm in all four operand layouts with padded leading
dimensions, bias / accumulate / relu / split-K / batch; tatt_conv_fwd + dgrad + wgrad with odd image sizes and channel counts)
against fp64 torch.  The fixed shapes of the network exercise a handful of dispatch paths (FAST loaders, packed weights); this
walks the rest: ragged tiles, unaligned strides, K tails, tiny and empty-ish problems."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.util import rel_l2

pytestmark = pytest.mark.gpu


def L():
    from tatt_amd.lib import lib
    return lib()


def ws(n=16 << 20):
    return torch.empty(n, device="cuda")


@pytest.mark.parametrize("seed", range(4))
def test_gemm_random_shapes_and_layouts(seed):
    rng = np.random.RandomState(100 + seed)
    g = torch.Generator().manual_seed(seed)
    W = ws()
    for case in range(30):
        M, N, K = int(rng.choice([1, 7, 33, 64, 130, 300, 1000])), int(rng.choice([1, 4, 40, 64, 96, 130])), int(rng.choice([1, 5, 32, 37, 64, 100, 257]))
        ta, tb = bool(rng.randint(2)), bool(rng.randint(2))                     # A stored [K][M] / B stored [K][N]
        pa, pb, pc = (int(rng.choice([0, 1, 4, 12])) for _ in range(3))         # leading-dimension padding
        bias, acc, relu = bool(rng.randint(2)), bool(rng.randint(2)), bool(rng.randint(2))
        sk = int(rng.choice([1, 1, 2, 5])) if K >= 64 else 1
        A = torch.randn(*((K, M + pa) if ta else (M, K + pa)), generator=g)
        B = torch.randn(*((K, N + pb) if tb else (N, K + pb)), generator=g)
        C0 = torch.randn(M, N + pc, generator=g)
        bv = torch.randn(N, generator=g)
        Am = A[:, :M].t() if ta else A[:, :K]
        Bm = B[:, :N].t() if tb else B[:, :K]
        ref = Am.double() @ Bm.double().t()
        if bias:
            ref = ref + bv.double()
        if acc:
            ref = ref + C0[:, :N].double()
        if relu:
            ref = torch.relu(ref)
        Cg = C0.cuda()
        sam, sak = (1, M + pa) if ta else (K + pa, 1)
        sbn, sbk = (1, N + pb) if tb else (K + pb, 1)
        L().gemm(A.cuda(), B.cuda(), Cg, bv.cuda() if bias else None, M, N, K, sam, sak, sbn, sbk, N + pc, 1, 0, 0, 0, int(acc), int(relu),
                 sk, W if sk > 1 else None)
        tag = f"case {case}: M{M} N{N} K{K} ta{ta} tb{tb} pad{pa},{pb},{pc} bias{bias} acc{acc} relu{relu} sk{sk}"
        scale = max(1.0, float(ref.abs().max()))
        assert float((Cg[:, :N].cpu().double() - ref).abs().max()) <= 3e-5 * scale * math.sqrt(max(K, 1)), tag
        assert torch.equal(Cg[:, N:].cpu(), C0[:, N:]), tag + " (wrote into the padding)"


@pytest.mark.parametrize("seed", range(3))
def test_conv_random_shapes(seed):
    rng = np.random.RandomState(200 + seed)
    g = torch.Generator().manual_seed(seed)
    W = ws()
    for case in range(8):
        B, H, Wd = int(rng.choice([1, 2, 3])), int(rng.choice([1, 2, 5, 8, 16])), int(rng.choice([1, 3, 8, 26, 32, 64]))
        Cin, Cout = int(rng.choice([4, 8, 32, 36, 64, 96])), int(rng.choice([4, 12, 32, 64, 100]))
        k = int(rng.choice([1, 3, 3, 9]))
        x = torch.randn(B, Cin, H, Wd, generator=g).double().requires_grad_(True)
        w = (torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).double().requires_grad_(True)
        b = torch.randn(Cout, generator=g).double()
        dy = torch.randn(B, Cout, H, Wd, generator=g).double()
        y = F.conv2d(x, w, b, padding=k // 2)
        gx, gw = torch.autograd.grad(y, (x, w), dy)
        xn = x.detach().float().permute(0, 2, 3, 1).contiguous().cuda()
        dyn = dy.float().permute(0, 2, 3, 1).contiguous().cuda()
        wf = w.detach().float().cuda()
        T = k * k
        tag = f"case {case}: B{B} H{H} W{Wd} Cin{Cin} Cout{Cout} k{k}"
        out = torch.empty(B, H, Wd, Cout, device="cuda")
        L().conv_fwd(xn, wf, b.float().cuda(), out, B, H, Wd, Cin, Cout, k, k, Cin, Cout, Cin * T, T, 0, 1, 0, 0)
        assert rel_l2(out.permute(0, 3, 1, 2).cpu().double().numpy(), y.detach().numpy()) < 2e-5, tag + " fwd"
        dx = torch.empty(B, H, Wd, Cin, device="cuda")
        L().conv_fwd(dyn, wf, None, dx, B, H, Wd, Cout, Cin, k, k, Cout, Cin, T, Cin * T, T - 1, -1, 0, 0)
        assert rel_l2(dx.permute(0, 3, 1, 2).cpu().double().numpy(), gx.numpy()) < 2e-5, tag + " dgrad"
        pf, pd = torch.empty(wf.numel(), device="cuda"), torch.empty(wf.numel(), device="cuda")
        L().conv_pack_w(wf, pf, pd, Cout, Cin, k, k)
        out2 = torch.empty_like(out)
        L().conv_fwd_splitk(xn, pf, b.float().cuda(), out2, B, H, Wd, Cin, Cout, k, k, Cin, Cout, T * Cin, 1, 0, Cin, 0, 0, 2, W)
        assert rel_l2(out2.permute(0, 3, 1, 2).cpu().double().numpy(), y.detach().numpy()) < 2e-5, tag + " fwd packed split-K"
        sk = int(rng.choice([1, 3]))
        dW = torch.zeros(Cout, Cin, k, k, device="cuda")
        L().conv_wgrad(xn, dyn, dW, B, H, Wd, Cin, Cout, k, k, Cin, Cout, sk, W, 1)
        assert rel_l2(dW.cpu().double().numpy(), gw.numpy()) < 3e-5, tag + f" wgrad sk{sk}"
