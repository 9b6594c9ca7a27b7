"""TATT_F32X3 (flag 4 of the GEMM / convolution entry points, model.matmul_precision = "f32x3"): fp32 products formed on the bf16
matrix core from an exact three-piece bf16 split of every operand element (six terms, fp32 accumulate).  Opt-in; the default path
multiplies on fp32 MFMA.  Bar: the same as the fp32 kernels' -- the result must be as close to the fp64 product of the UNROUNDED
fp32 operands as the fp32 FMA chain is (no worse than 1.5x the fp32 path's own error, which is 2e-7 .. 2e-6 with the length of the sum).  Model-level pins:
tests/test_gpu_model.py runs the B=64 reference golden and the eval-mode gradient pin in this mode with unchanged tolerances."""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.util import rel_l2

pytestmark = pytest.mark.gpu
X3 = 4


def L():
    from tatt_amd.lib import lib
    return lib()


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def err(got, ref):
    assert torch.isfinite(got).all()
    return rel_l2(got.detach().cpu().double().numpy(), ref.detach().double().numpy())


def both(run, ref, what):
    """run(flags) -> tensor comparable with ref (fp64)"""
    e32, e3 = err(run(0), ref), err(run(X3), ref)
    # 2.5e-6: what ONE fp32 FMA chain per output leaves over the 65,536-row weight-gradient sums (2.2e-6 with the generic tile); the
    # fp32 wgrad192 kernel sums two half-chains per output and is itself below that (1.6e-6)
    assert (e3 <= 1.5 * e32 + 2e-8 or e3 <= 2.5e-6) and e3 <= 5e-6, f"{what}: f32x3 rel-L2 {e3:.3e}, fp32 MFMA {e32:.3e}"
    return e32, e3


@pytest.mark.parametrize("M,N,K", [(1000, 96, 64), (4096, 64, 2448), (132, 40, 96), (65536, 192, 128)])
def test_gemm_nt(M, N, K):
    A, B_, bias = rnd(M, K), rnd(N, K, seed=1), rnd(N, seed=2)
    Ad, Bd, bd = A.cuda(), B_.cuda(), bias.cuda()
    ref = A.double() @ B_.double().t() + bias.double()

    def run(fl):
        out = torch.empty(M, N, device="cuda")
        L().gemm(Ad, Bd, out, bd, M, N, K, K, 1, K, 1, N, 1, 0, 0, 0, 0, fl, 1, None)
        return out
    both(run, ref, "gemm nt")
    Bt, At = B_.t().contiguous().cuda(), A.t().contiguous().cuda()
    ref0 = A.double() @ B_.double().t()
    wsp = torch.empty(4 * M * N, device="cuda")

    def nn(fl):                                               # B row-contiguous (the dX products)
        out = torch.empty(M, N, device="cuda")
        L().gemm(Ad, Bt, out, None, M, N, K, K, 1, 1, N, N, 1, 0, 0, 0, 0, fl, 1, None)
        return out

    def tn(fl):                                               # A row-contiguous, split-K (the dW products)
        out = torch.empty(M, N, device="cuda")
        L().gemm(At, Bd, out, None, M, N, K, 1, M, K, 1, N, 1, 0, 0, 0, 0, fl, 2, wsp)
        return out

    def tt(fl):
        out = torch.empty(M, N, device="cuda")
        L().gemm(At, Bt, out, None, M, N, K, 1, M, 1, N, N, 1, 0, 0, 0, 0, fl, 1, None)
        return out
    if M % 4 == 0 and N % 4 == 0:                             # the FAST loaders (and with them the split kernels) need 16-byte rows
        both(nn, ref0, "gemm nn")
        both(tn, ref0, "gemm tn split-K")
        both(tt, ref0, "gemm tt")
    # values with a wide dynamic range inside one dot product: the split must stay exact piece by piece
    A2 = A * torch.exp(rnd(M, K, seed=5) * 4)
    ref2 = A2.double() @ B_.double().t()
    A2d = A2.cuda()

    def run2(fl):
        out = torch.empty(M, N, device="cuda")
        L().gemm(A2d, Bd, out, None, M, N, K, K, 1, K, 1, N, 1, 0, 0, 0, 0, fl, 1, None)
        return out
    both(run2, ref2, "gemm nt, wide dynamic range")


# (2,16,64,..) / (64,16,64,..): 64-wide maps with an even number of rows take the LDS-halo kernel; (2,3,64,..) (odd H) and (1,2,64,..)
# take the tap-unrolled per-tap loader; the others the generic loop
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 16, 64, 64, 64), (2, 16, 64, 64, 256), (3, 8, 32, 32, 64), (5, 4, 8, 128, 256), (64, 16, 64, 64, 64),
                                            (2, 3, 64, 64, 64), (1, 2, 64, 64, 64), (3, 4, 64, 64, 64)])
def test_conv3x3_fwd_dgrad(B, H, W, Cin, Cout):
    x, dy = rnd(B, Cin, H, W), rnd(B, Cout, H, W, seed=3)
    w, b = rnd(Cout, Cin, 3, 3, seed=1) / math.sqrt(Cin * 9), rnd(Cout, seed=2)
    y = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    gx = F.conv_transpose2d(dy.double(), w.double(), padding=1)
    xn, dyn = x.permute(0, 2, 3, 1).contiguous().cuda(), dy.permute(0, 2, 3, 1).contiguous().cuda()
    wf, wd = torch.empty(w.numel(), device="cuda"), torch.empty(w.numel(), device="cuda")
    L().conv_pack_w(w.cuda(), wf, wd, Cout, Cin, 3, 3)
    bd = b.cuda()

    def fwd(fl):
        out = torch.empty(B, H, W, Cout, device="cuda")
        L().conv_fwd(xn, wf, bd, out, B, H, W, Cin, Cout, 3, 3, Cin, Cout, 9 * Cin, 1, 0, Cin, 0, fl)
        return out.permute(0, 3, 1, 2)

    def dgrad(fl):
        dx = torch.empty(B, H, W, Cin, device="cuda")
        L().conv_fwd(dyn, wd, None, dx, B, H, W, Cout, Cin, 3, 3, Cout, Cin, 9 * Cout, 1, 0, Cout, 0, fl)
        return dx.permute(0, 3, 1, 2)
    both(fwd, y, "conv fwd")
    both(dgrad, gx, "conv dgrad")
    wd64 = w.double().requires_grad_(True)
    gw, = torch.autograd.grad(F.conv2d(x.double(), wd64, None, padding=1), (wd64,), dy.double())
    wsp = torch.empty(8 << 20, device="cuda")

    def wgrad(fl):
        dW = torch.zeros(Cout, Cin, 3, 3, device="cuda")
        L().conv_wgrad(xn, dyn, dW, B, H, W, Cin, Cout, 3, 3, Cin, Cout, 4, wsp, 1 | fl)
        return dW
    both(wgrad, gw, "conv wgrad")


@pytest.mark.parametrize("act", [0, 2])
def test_conv_bn_epilogues(act):
    """EPI 1 (forward statistics) and EPI 2 (BatchNorm-backward sums + activation derivative) on top of the split products: the
    tile values and the per-tile partial sums must agree with the fp32 MFMA launch of the same call to fp32 rounding."""
    B, H, W, C = 4, 16, 64, 64
    R = B * H * W
    x = rnd(B, H, W, C).cuda()
    w = (rnd(C, C, 3, 3, seed=1) / 24).cuda()
    wf, wd = torch.empty(w.numel(), device="cuda"), torch.empty(w.numel(), device="cuda")
    L().conv_pack_w(w, wf, wd, C, C, 3, 3)
    bias = rnd(C, seed=2).cuda()
    nblk = -(-R // 128)
    outs = {}
    for fl in (0, X3):
        y, part = torch.empty(R, C, device="cuda"), torch.empty(nblk, 2, C, device="cuda")
        L().conv_fwd_bn(x, wf, bias, y, B, H, W, C, C, 3, 3, C, C, 9 * C, 1, 0, C, fl, 1, None, C, None, None, None, 0, part)
        bnx = rnd(R, C, seed=7).cuda()
        stat = torch.stack([rnd(C, seed=8) * 0.1, 1 + 0.1 * rnd(C, seed=9).abs()]).cuda()
        ga, be = (1 + 0.1 * rnd(C, seed=10)).cuda(), (0.1 * rnd(C, seed=11)).cuda()
        g, part2 = torch.empty(R, C, device="cuda"), torch.empty(nblk, 2, C, device="cuda")
        L().conv_fwd_bn(x, wd, None, g, B, H, W, C, C, 3, 3, C, C, 9 * C, 1, 0, C, fl, 2, bnx, C, stat, ga, be, act, part2)
        outs[fl] = (y, part, g, part2)
    for a, b, what in zip(outs[0], outs[X3], ("y", "stats partial", "g", "bwd partial")):
        r = rel_l2(b.cpu().double().numpy(), a.cpu().double().numpy())
        assert r <= 2e-6, f"{what}: {r:.3e}"


def test_flag_is_per_call_and_bf16_wins():
    M, N, K = 256, 64, 64
    A, B_ = rnd(M, K).cuda(), rnd(N, K, seed=1).cuda()
    ref = A.double().cpu() @ B_.double().cpu().t()
    o = [torch.empty(M, N, device="cuda") for _ in range(3)]
    for out, fl in zip(o, (X3, 0, X3 | 2)):
        L().gemm(A, B_, out, None, M, N, K, K, 1, K, 1, N, 1, 0, 0, 0, 0, fl, 1, None)
    assert err(o[0], ref) < 1e-6 and err(o[1], ref) < 1e-6
    assert err(o[2], ref) > 1e-4          # TATT_BF16 set as well: operands rounded to bf16
