"""Per-kernel parity: every C-ABI entry point against a plain torch-CPU fp32/fp64 statement of the same op
(or the oracle's restatement), on seeded inputs.  GPU only; calls go through the C-ABI (tatt_amd.lib)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tatt_ref as R
from tests.util import rel_l2

pytestmark = pytest.mark.gpu

TOL = 2e-5   # fp32 MFMA = exact fp32 fma chains; only the summation order differs from the CPU reference


def L():
    from tatt_amd.lib import lib
    return lib()


def cu(t):
    return t.contiguous().cuda()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).float()


def close(got, ref, tol=TOL, what=""):
    r = rel_l2(got.detach().cpu().double().numpy(), ref.detach().double().numpy())
    assert r <= tol, f"{what}: rel-L2 {r:.3e} > {tol}"
    assert torch.isfinite(got).all(), what


def ws(n=8 << 20):
    return torch.empty(n, device="cuda")


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(300, 64, 37), (1000, 96, 64), (64, 1536, 512), (52, 64, 2448), (7, 40, 512), (128, 64, 32)])
def test_gemm_nt_bias_relu_acc(M, N, K):
    A, B_, bias, C0 = rnd(M, K), rnd(N, K, seed=1), rnd(N, seed=2), rnd(M, N, seed=3)
    ref = A.double() @ B_.double().t() + bias.double()
    out = torch.empty(M, N, device="cuda")
    L().gemm(cu(A), cu(B_), out, cu(bias), M, N, K, K, 1, K, 1, N, 1, 0, 0, 0, 0, 0, 1, None)
    close(out, ref, what="nt")
    out2 = cu(C0)
    L().gemm(cu(A), cu(B_), out2, cu(bias), M, N, K, K, 1, K, 1, N, 1, 0, 0, 0, 1, 1, 1, None)
    close(out2, torch.relu(ref + C0.double()), what="nt+acc+relu")


@pytest.mark.parametrize("M", [1024, 4099, 65536])
@pytest.mark.parametrize("N,K", [(64, 64), (192, 64), (64, 192)])
def test_gemm_tall_narrow_shapes(M, N, K):
    """The tall K,N in {64,192} linears of the SRBs / TP decoder as the engine issues them: NT and NN forms, bias / relu /
    accumulate, leading dimensions wider than the payload (ldw=128 halves of a stacked weight, ldc=192 slices), ragged M."""
    A, bias, C0 = rnd(M, K), rnd(N, seed=2), rnd(M, N, seed=3)
    Wb = rnd(N, 2 * K, seed=1)                      # the operand is the right half of a [N, 2K] matrix (ldw = 2K)
    W = Wb[:, K:]
    ref = A.double() @ W.double().t() + bias.double()
    out = torch.empty(M, N, device="cuda")
    L().gemm(cu(A), cu(Wb)[:, K:], out, cu(bias), M, N, K, K, 1, 2 * K, 1, N, 1, 0, 0, 0, 0, 0, 1, None)
    close(out, ref, what="nt")
    out2 = cu(C0)
    L().gemm(cu(A), cu(Wb)[:, K:], out2, cu(bias), M, N, K, K, 1, 2 * K, 1, N, 1, 0, 0, 0, 1, 1, 1, None)
    close(out2, torch.relu(ref + C0.double()), what="nt+acc+relu")
    # wide destination and source rows: C is columns [N, 2N) of a [M, 3N] buffer, A is columns [K, 2K) of [M, 2K]
    Ab, Cb = rnd(M, 2 * K, seed=4), torch.full((M, 3 * N), 7.0, device="cuda")
    Ad = cu(Ab)
    L().gemm(Ad[:, K:], cu(W).contiguous(), Cb[:, N:], None, M, N, K, 2 * K, 1, K, 1, 3 * N, 1, 0, 0, 0, 0, 0, 1, None)
    close(Cb[:, N:2 * N], Ab[:, K:].double() @ W.double().t(), what="nt wide")
    assert (Cb[:, :N] == 7).all() and (Cb[:, 2 * N:] == 7).all()
    # NN: dx[M,N] = dy[M,K] @ V[K,N]  (B strides sbn=1, sbk=ldv)
    V = rnd(K, N, seed=5)
    dx = cu(C0)
    L().gemm(cu(A), cu(V), dx, None, M, N, K, K, 1, 1, N, N, 1, 0, 0, 0, 1, 0, 1, None)
    close(dx, C0.double() + A.double() @ V.double(), what="nn+acc")


def test_gemm_nn_tn_splitk_batched():
    R_, N, K = 4100, 96, 64
    dy, W, x = rnd(R_, N), rnd(N, K, seed=1), rnd(R_, K, seed=2)
    dx = torch.empty(R_, K, device="cuda")
    L().gemm(cu(dy), cu(W), dx, None, R_, K, N, N, 1, 1, K, K, 1, 0, 0, 0, 0, 0, 1, None)      # dx = dy @ W
    close(dx, dy.double() @ W.double(), what="nn")
    for sk in (1, 5, 16):
        dW = cu(torch.ones(N, K))
        L().gemm(cu(dy), cu(x), dW, None, N, K, R_, 1, N, 1, K, K, 1, 0, 0, 0, 1, 0, sk, ws())   # dW += dy^T x
        close(dW, 1 + dy.double().t() @ x.double(), what=f"tn splitk {sk}")
    # strided sub-blocks (ld > width), as the GRU gradient GEMMs use them
    big = rnd(R_, 192, seed=5)
    hp = rnd(R_, 64, seed=6)
    dW = cu(torch.zeros(96, 32))
    L().gemm(cu(big)[:, 96:], cu(hp)[:, 32:], dW, None, 96, 32, R_, 1, 192, 1, 64, 32, 1, 0, 0, 0, 1, 0, 8, ws())
    close(dW, big[:, 96:].double().t() @ hp[:, 32:].double(), what="tn strided")
    # batched (the RPE recurrence step)
    h, Wh = rnd(2, 64, 512, seed=7), rnd(2, 1536, 512, seed=8, scale=0.05)
    gh = torch.empty(2, 64, 1536, device="cuda")
    L().gemm(cu(h), cu(Wh), gh, None, 64, 1536, 512, 512, 1, 512, 1, 1536, 2, 64 * 512, 1536 * 512, 64 * 1536, 0, 0, 1, None)
    close(gh, torch.einsum("bmk,bnk->bmn", h.double(), Wh.double()), what="batched")


CONVS = [(2, 16, 64, 64, 64, 3), (2, 16, 64, 4, 64, 9), (1, 32, 128, 64, 4, 9), (3, 8, 32, 32, 64, 3),
         (2, 1, 2, 256, 256, 3), (2, 16, 64, 64, 256, 3), (5, 2, 8, 128, 256, 3), (4, 2, 8, 128, 256, 3),
         (4, 4, 16, 64, 128, 3), (4, 1, 4, 256, 256, 3)]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k", CONVS)
def test_conv_fwd_dgrad_wgrad(B, H, W, Cin, Cout, k):
    x = rnd(B, Cin, H, W).requires_grad_(True)
    w = (rnd(Cout, Cin, k, k, seed=1) / math.sqrt(Cin * k * k)).requires_grad_(True)
    b = rnd(Cout, seed=2).requires_grad_(True)
    y = F.conv2d(x.double(), w.double(), b.double(), padding=k // 2)
    dy = rnd(B, Cout, H, W, seed=3)
    y.backward(dy.double())
    xn, dyn = cu(x.detach().permute(0, 2, 3, 1)), cu(dy.permute(0, 2, 3, 1))
    T = k * k
    out = torch.empty(B, H, W, Cout, device="cuda")
    L().conv_fwd(xn, cu(w.detach()), cu(b.detach()), out, B, H, W, Cin, Cout, k, k, Cin, Cout, Cin * T, T, 0, 1, 0, 0)
    close(out.permute(0, 3, 1, 2), y, what="fwd")
    dx = torch.empty(B, H, W, Cin, device="cuda")
    L().conv_fwd(dyn, cu(w.detach()), None, dx, B, H, W, Cout, Cin, k, k, Cout, Cin, T, Cin * T, T - 1, -1, 0, 0)
    close(dx.permute(0, 3, 1, 2), x.grad, what="dgrad")
    # the per-step packed layouts the engine uses: bitwise the same sums in the same order
    wf, wd = torch.empty(w.numel(), device="cuda"), torch.empty(w.numel(), device="cuda")
    L().conv_pack_w(cu(w.detach()), wf, wd, Cout, Cin, k, k)
    out2, dx2 = torch.empty_like(out), torch.empty_like(dx)
    L().conv_fwd(xn, wf, cu(b.detach()), out2, B, H, W, Cin, Cout, k, k, Cin, Cout, T * Cin, 1, 0, Cin, 0, 0)
    L().conv_fwd(dyn, wd, None, dx2, B, H, W, Cout, Cin, k, k, Cout, Cin, T * Cout, 1, 0, Cout, 0, 0)
    assert torch.equal(out2, out) and torch.equal(dx2, dx)
    # split-K variant (small feature maps of the STN head), with relu and accumulate in the reduce kernel
    out3 = cu(torch.ones(B, H, W, Cout))
    L().conv_fwd_splitk(xn, wf, cu(b.detach()), out3, B, H, W, Cin, Cout, k, k, Cin, Cout, T * Cin, 1, 0, Cin, 1, 1, 3, ws())
    close(out3.permute(0, 3, 1, 2), torch.relu(y + 1), what="fwd splitk+acc+relu")
    for sk in (1, 4):
        dW = torch.zeros(Cout, Cin, k, k, device="cuda")
        L().conv_wgrad(xn, dyn, dW, B, H, W, Cin, Cout, k, k, Cin, Cout, sk, ws(), 1)
        close(dW, w.grad, what=f"wgrad sk={sk}")
    db = torch.zeros(Cout, device="cuda")
    L().colsum(dyn, Cout, B * H * W, Cout, db, 1, ws())
    close(db, b.grad, what="dbias")


@pytest.mark.parametrize("B,H,W,sk", [(3, 16, 64, 7), (64, 16, 64, 256), (1, 5, 6, 3), (2, 3, 2, 1), (1, 1, 2, 5), (5, 16, 64, 1000)])
def test_conv3x3_wgrad_borders_and_slices(B, H, W, sk):
    """3x3, 64 -> 64 weight gradient (the SRB shape): image borders for every tap, split counts that do not divide the pixel
    count, accumulate on and off, operands that are column slices of wider NHWC buffers."""
    Cin = Cout = 64
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + W)
    x = torch.randn(B, Cin, H, W, generator=g)
    dy = torch.randn(B, Cout, H, W, generator=g)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x.double(), w, padding=1).backward(dy.double())
    xb = torch.full((B, H, W, 2 * Cin), 5.0)                      # operands are column slices of wider NHWC buffers
    xb[..., Cin:] = x.permute(0, 2, 3, 1)
    dyb = torch.full((B, H, W, 3 * Cout), -3.0)
    dyb[..., Cout:2 * Cout] = dy.permute(0, 2, 3, 1)
    xn, dyn = cu(xb), cu(dyb)
    dW = torch.full((Cout, Cin, 3, 3), 2.0, device="cuda")
    L().conv_wgrad(xn.view(-1, 2 * Cin)[:, Cin:], dyn.view(-1, 3 * Cout)[:, Cout:], dW, B, H, W, Cin, Cout, 3, 3, 2 * Cin, 3 * Cout, sk, ws(48 << 20), 1)
    close(dW, w.grad + 2.0, what="accumulate")
    L().conv_wgrad(xn.view(-1, 2 * Cin)[:, Cin:], dyn.view(-1, 3 * Cout)[:, Cout:], dW, B, H, W, Cin, Cout, 3, 3, 2 * Cin, 3 * Cout, sk, ws(48 << 20), 0)
    close(dW, w.grad, what="overwrite")


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,sk", [(2, 4, 32, 64, 48, 3, 3), (1, 2, 96, 128, 20, 3, 5), (3, 3, 32, 192, 64, 1, 2), (1, 16, 64, 64, 64, 3, 85),
                                                 (2, 5, 64, 64, 36, 3, 1)])
def test_conv_wgrad_192_row_tile(B, H, W, Cin, Cout, k, sk):
    """Shapes that take wgrad192_kernel (fp32, T*Cin a multiple of 192, W a multiple of 32, 16 < Cout <= 64): partial N tile,
    several M tiles, 1x1, odd tile counts per slab, slabs past the end of the pixels."""
    T = k * k
    assert (T * Cin) % 192 == 0 and W % 32 == 0
    x, dy = rnd(B, Cin, H, W), rnd(B, Cout, H, W, seed=3)
    w = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, padding=k // 2).backward(dy.double())
    xn, dyn = cu(x.permute(0, 2, 3, 1)), cu(dy.permute(0, 2, 3, 1))
    dW = torch.full((Cout, Cin, k, k), 1.0, device="cuda")
    L().conv_wgrad(xn, dyn, dW, B, H, W, Cin, Cout, k, k, Cin, Cout, sk, ws(), 1)
    close(dW, w.grad + 1.0, what="accumulate")
    L().conv_wgrad(xn, dyn, dW, B, H, W, Cin, Cout, k, k, Cin, Cout, sk, ws(), 0)
    close(dW, w.grad, what="overwrite")


@pytest.mark.parametrize("B,H,W,Cbig,Csmall,k", [(2, 32, 128, 64, 4, 9), (3, 16, 64, 64, 4, 9), (2, 7, 70, 32, 4, 9), (1, 3, 5, 16, 4, 9), (2, 8, 8, 32, 8, 3), (1, 5, 7, 64, 16, 3)])
def test_conv_skinny_fwd_and_dgrad(B, H, W, Cbig, Csmall, k):
    """Skinny-Cout path: forward Cbig->Csmall, and the dgrad of a Csmall->Cbig conv (model/tsrn.py:625, :598)."""
    T = k * k
    x = rnd(B, Cbig, H, W)
    w = rnd(Csmall, Cbig, k, k, seed=1) / math.sqrt(Cbig * T)
    b = rnd(Csmall, seed=2)
    y = F.conv2d(x.double(), w.double(), b.double(), padding=k // 2)
    wsz = L().conv_skinny_workspace(B, H, W, Cbig, Csmall, k, k)
    out = torch.empty(B, H, W, Csmall, device="cuda")
    L().conv_skinny_fwd(cu(x.permute(0, 2, 3, 1)), cu(w), cu(b), out, B, H, W, Cbig, Csmall, k, k, Cbig, Csmall, Cbig * T, T, 0, 1, 0, ws(wsz))
    close(out.permute(0, 3, 1, 2), y, what="skinny fwd")
    # dgrad of conv(Csmall -> Cbig): dx[Csmall] = conv^T(dy[Cbig])
    xs = rnd(B, Csmall, H, W, seed=3).double().requires_grad_(True)
    w2 = (rnd(Cbig, Csmall, k, k, seed=4) / math.sqrt(Csmall * T)).double()
    dy = rnd(B, Cbig, H, W, seed=5)
    F.conv2d(xs, w2, None, padding=k // 2).backward(dy.double())
    dx = cu(torch.ones(B, H, W, Csmall))
    L().conv_skinny_fwd(cu(dy.permute(0, 2, 3, 1)), cu(w2.float()), None, dx, B, H, W, Cbig, Csmall, k, k, Cbig, Csmall, T, Csmall * T,
                        T - 1, -1, 1, ws(wsz))
    close(dx.permute(0, 3, 1, 2), 1 + xs.grad, what="skinny dgrad (+accumulate)")


@pytest.mark.parametrize("R_,M,N,lddy,ldx", [(65536, 64, 64, 64, 64), (4100, 96, 32, 192, 64), (1664, 64, 37, 64, 37), (52, 96, 64, 192, 64),
                                             (3000, 64, 64, 64, 128), (7, 40, 20, 40, 20)])
def test_tn_small_dw_and_bias(R_, M, N, lddy, ldx):
    dy, x = rnd(R_, lddy), rnd(R_, ldx, seed=1)
    dW0, db0 = rnd(M, N + 3, seed=2), rnd(M, seed=3)
    dW, db = cu(dW0), cu(db0)
    L().tn_small(cu(dy), lddy, cu(x), ldx, dW, N + 3, db, M, N, R_, ws())
    ref = dW0.double().clone()
    ref[:, :N] += dy[:, :M].double().t() @ x[:, :N].double()
    close(dW, ref, what="tn_small dW")
    close(db, db0.double() + dy[:, :M].double().sum(0), 1e-5 if R_ < 60000 else 1e-4, "tn_small db")


def test_gemm_batched_splitk():
    dgh, Wh = rnd(2, 64, 1536, seed=1) * 0.1, rnd(2, 1536, 512, seed=2) * 0.05
    carry0 = rnd(2, 64, 512, seed=3)
    carry = cu(carry0)
    L().gemm(cu(dgh), cu(Wh), carry, None, 64, 512, 1536, 1536, 1, 1, 512, 512, 2, 64 * 1536, 1536 * 512, 64 * 512, 1, 0, 8, ws())
    close(carry, carry0.double() + torch.einsum("bmk,bkn->bmn", dgh.double(), Wh.double()), what="batched split-K NN + accumulate")


@pytest.mark.parametrize("R_,C,act", [(2048, 64, 2), (100, 32, 1), (4, 512, 1), (20000, 256, 0), (4, 256, 1), (256, 128, 1),
                                       (1024, 64, 1), (64, 256, 1), (16, 256, 1), (312, 512, 1), (600, 256, 1), (78, 512, 1)])
def test_batchnorm_train_fwd_bwd(R_, C, act):
    x = (rnd(R_, C) * 2 + 0.5).requires_grad_(True)
    g_, b_ = (1 + 0.2 * rnd(C, seed=1)).requires_grad_(True), rnd(C, seed=2).requires_grad_(True)
    rm, rv = rnd(C, seed=3) * 0.1, 1 + 0.3 * torch.rand(C)
    rm_ref, rv_ref = rm.clone().double(), rv.clone().double()
    y = F.batch_norm(x.double(), rm_ref, rv_ref, g_.double(), b_.double(), True, 0.1, 1e-5)
    fa = {0: lambda t: t, 1: torch.relu, 2: R.mish}[act]
    ya = fa(y)
    dy = rnd(R_, C, seed=4)
    ya.backward(dy.double())
    stat = torch.empty(2, C, device="cuda")
    rmd, rvd = cu(rm), cu(rv)
    w = ws()
    L().bn_stats(cu(x.detach()), C, R_, C, 1e-5, 0.1, stat, rmd, rvd, w)
    out = torch.empty(R_, C, device="cuda")
    L().bn_apply(cu(x.detach()), C, stat, cu(g_.detach()), cu(b_.detach()), None, C, out, C, R_, C, act)
    close(out, ya, what="bn fwd")
    close(rmd, rm_ref, 1e-5, "running_mean")
    close(rvd, rv_ref, 1e-5, "running_var")
    dx, dg, db = torch.empty(R_, C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    L().bn_bwd(cu(x.detach()), C, cu(dy), C, stat, cu(g_.detach()), cu(b_.detach()), dx, C, dg, db, R_, C, act, 1, 1, 0, w)
    tol = 2e-4 if R_ <= 4 else 5e-5          # 4-row batches: xhat = +-1, differences of nearly equal numbers
    close(dx, x.grad, tol, "bn dx")
    close(dg, g_.grad, tol, "bn dgamma")
    close(db, b_.grad, tol, "bn dbeta")


@pytest.mark.parametrize("rows", [1000, 1, 19, 1664])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_layernorm_residual_dropout(p, rows):
    x, res = rnd(rows, 64).requires_grad_(True), rnd(rows, 64, seed=1).requires_grad_(True)
    g_, b_ = (1 + 0.2 * rnd(64, seed=2)).requires_grad_(True), rnd(64, seed=3).requires_grad_(True)
    drop = R.HashDropout(p, 7)
    mask = drop.mask("enc_drop1", (rows, 64)) if p > 0 else torch.ones(rows, 64)
    v = x.double() + res.double() * mask.double()
    y = F.layer_norm(v, (64,), g_.double(), b_.double())
    dy = rnd(rows, 64, seed=4)
    y.backward(0.5 * dy.double())
    ss = (7 * 0x85EBCA77 + R.DROP_SITES["enc_drop1"] * 0xC2B2AE3D) & 0xFFFFFFFF
    cfg = (ss, int(p * (1 << 24)), 1 / (1 - p), None) if p > 0 else (0, 0, 1.0, None)
    vout, out = torch.empty(rows, 64, device="cuda"), torch.empty(rows, 64, device="cuda")
    L().ln_fwd(cu(x.detach()), cu(res.detach()), *cfg, cu(g_.detach()), cu(b_.detach()), vout, out, rows, 1e-5)
    close(out, y, what="ln fwd")
    close(vout, v, what="ln v")
    dv, dres = torch.empty(rows, 64, device="cuda"), torch.empty(rows, 64, device="cuda")
    dg, db = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda")
    L().ln_bwd(vout, cu(dy), 0.5, cu(g_.detach()), dv, 0, dres, *cfg, dg, db, 1, rows, 1e-5, ws())
    close(dv, x.grad, 5e-5, "ln dx")
    close(dres, res.grad, 5e-5, "ln dres")
    close(dg, g_.grad, 5e-5, "ln dgamma")
    close(db, b_.grad, 5e-5, "ln dbeta")


@pytest.mark.parametrize("vertical", [False, True])
def test_gru32_fwd_bwd(vertical):
    B, H, W = 2, 16, 64
    P = {}
    for sfx in ("", "_reverse"):
        P["weight_ih_l0" + sfx] = rnd(96, 64, seed=len(sfx)) * 0.2
        P["weight_hh_l0" + sfx] = rnd(96, 32, seed=1 + len(sfx)) * 0.3
        P["bias_ih_l0" + sfx] = rnd(96, seed=2 + len(sfx)) * 0.1
        P["bias_hh_l0" + sfx] = rnd(96, seed=3 + len(sfx)) * 0.1
    P = {k: v.double().requires_grad_(True) for k, v in P.items()}
    x = rnd(B, H, W, 64, seed=9).double().requires_grad_(True)             # NHWC
    seq = x.permute(0, 2, 1, 3).reshape(B * W, H, 64) if vertical else x.reshape(B * H, W, 64)
    o = R.bigru(P, "", seq)
    o = o.view(B, W, H, 64).permute(0, 2, 1, 3) if vertical else o.view(B, H, W, 64)
    dout = rnd(B, H, W, 64, seed=10)
    o.backward(dout.double())
    Rr = B * H * W
    gi = torch.empty(Rr, 192, device="cuda")
    xd = cu(x.detach().float().reshape(Rr, 64))
    f32 = lambda k: cu(P[k].detach().float())
    for d, sfx in enumerate(("", "_reverse")):
        L().gemm(xd, f32("weight_ih_l0" + sfx), gi.view(-1)[96 * d:], f32("bias_ih_l0" + sfx), Rr, 96, 64, 64, 1, 64, 1, 192, 1, 0, 0,
                 0, 0, 0, 1, None)
    sm = (B * W, H, W, H * W, 1, W) if vertical else (B * H, W, 1, W, 0, 1)
    out, sv = torch.empty(Rr, 64, device="cuda"), torch.empty(Rr, 2, 4, 32, device="cuda")
    L().gru32_fwd(gi, f32("weight_hh_l0"), f32("weight_hh_l0_reverse"), f32("bias_hh_l0"), f32("bias_hh_l0_reverse"), out, sv, *sm)
    close(out.view(B, H, W, 64), o, what="gru fwd")
    # the variant that also writes res + out: same recurrence bit for bit, the sum exact
    res = cu(rnd(Rr, 64, seed=11))
    out_b, sv_b, out2 = torch.empty_like(out), torch.empty_like(sv), torch.empty_like(out)
    L().gru32_fwd_res(gi, f32("weight_hh_l0"), f32("weight_hh_l0_reverse"), f32("bias_hh_l0"), f32("bias_hh_l0_reverse"), out_b, sv_b, res, out2, *sm)
    assert torch.equal(out_b, out) and torch.equal(sv_b, sv) and torch.equal(out2, res + out)
    dgi, dghn, hp = torch.empty(Rr, 192, device="cuda"), torch.empty(Rr, 64, device="cuda"), torch.empty(Rr, 64, device="cuda")
    L().gru32_bwd(cu(dout.reshape(Rr, 64)), out, sv, f32("weight_hh_l0"), f32("weight_hh_l0_reverse"), dgi, dghn, hp, *sm)
    dgh = dgi.clone()                                                        # dgh = dgi with the n gate replaced by dghn (= dn * r)
    for d in range(2):
        dgh[:, 96 * d + 64:96 * d + 96] = dghn[:, 32 * d:32 * d + 32]
    dx = torch.zeros(Rr, 64, device="cuda")
    for d, sfx in enumerate(("", "_reverse")):
        dW = torch.zeros(96, 32, device="cuda")
        L().gemm(dgh.view(-1)[96 * d:], hp.view(-1)[32 * d:], dW, None, 96, 32, Rr, 1, 192, 1, 64, 32, 1, 0, 0, 0, 1, 0, 4, ws())
        close(dW, P["weight_hh_l0" + sfx].grad, 5e-5, "dW_hh" + sfx)
        dWi = torch.zeros(96, 64, device="cuda")
        L().gemm(dgi.view(-1)[96 * d:], xd, dWi, None, 96, 64, Rr, 1, 192, 1, 64, 64, 1, 0, 0, 0, 1, 0, 4, ws())
        close(dWi, P["weight_ih_l0" + sfx].grad, 5e-5, "dW_ih" + sfx)
        db = torch.zeros(96, device="cuda")
        L().colsum(dgh.view(-1)[96 * d:], 192, Rr, 96, db, 1, ws())
        close(db, P["bias_hh_l0" + sfx].grad, 5e-5, "db_hh" + sfx)
        L().gemm(dgi.view(-1)[96 * d:], f32("weight_ih_l0" + sfx), dx, None, Rr, 64, 96, 192, 1, 1, 64, 64, 1, 0, 0, 0, 1, 0, 1, None)
    close(dx.view(B, H, W, 64), x.grad, 5e-5, "gru dx")


@pytest.mark.parametrize("B,H,W,C,kh,kw", [(64, 4, 16, 128, 2, 2), (64, 2, 8, 256, 2, 2), (64, 1, 4, 256, 1, 2), (64, 1, 2, 256, 1, 1),
                                          (64, 1, 1, 512, 1, 1), (3, 2, 6, 24, 2, 2), (5, 1, 1, 8, 1, 1), (7, 3, 4, 16, 1, 2)])
def test_bn_pool_small_vs_torch_and_generic_launches(B, H, W, C, kh, kw):
    """tatt_bn_pool_small_fwd / _bwd (BatchNorm batch statistics + ReLU + max-pool of the STN head's small maps, model/stn_head.py:19-20,
    :38-44, :51-52, one launch per direction) against fp64 torch and against the generic launches they replace."""
    x = rnd(B, C, H, W, seed=1).double().requires_grad_(True)
    ga, be = (1 + 0.2 * rnd(C, seed=2)).double().requires_grad_(True), (0.1 * rnd(C, seed=3)).double().requires_grad_(True)
    rm0, rv0 = 0.1 * rnd(C, seed=4), 1 + 0.1 * rnd(C, seed=5).abs()
    rm, rv = rm0.double().clone(), rv0.double().clone()
    a = torch.relu(F.batch_norm(x, rm, rv, ga, be, True, 0.1, 1e-5))
    y = F.max_pool2d(a, (kh, kw), (kh, kw)) if kh * kw > 1 else a
    dy = rnd(*y.shape, seed=6)
    y.backward(dy.double())
    ref = lambda t: t.detach().permute(0, 2, 3, 1).contiguous()             # CPU fp64, NHWC
    nhwc = lambda t: ref(t).float().cuda()
    R, Rp = B * H * W, B * (H // kh) * (W // kw)
    xd, gad, bed, dyd = nhwc(x).view(R, C), cu(ga.detach().float()), cu(be.detach().float()), nhwc(dy).view(Rp, C)
    assert L().bn_pool_small_supported(B, H, W, C, kh, kw)
    stat, act, pooled = torch.empty(2, C, device="cuda"), torch.empty(R, C, device="cuda"), torch.empty(Rp, C, device="cuda")
    rmd, rvd = cu(rm0), cu(rv0)
    L().bn_pool_small_fwd(xd, stat, gad, bed, rmd, rvd, act, pooled if kh * kw > 1 else None, B, H, W, C, kh, kw, 1e-5, 0.1)
    close(act, ref(a).view(R, C), what="relu(bn(x))")
    if kh * kw > 1:
        close(pooled, ref(y).view(Rp, C), what="pooled")
    close(rmd, rm, what="running_mean")
    close(rvd, rv, what="running_var")
    dx, dg, db = torch.empty(R, C, device="cuda"), torch.ones(C, device="cuda"), torch.ones(C, device="cuda")
    L().bn_pool_small_bwd(xd, act, dyd, stat, gad, dx, dg, db, B, H, W, C, kh, kw)
    close(dx, ref(x.grad).view(R, C), 5e-5, "dx")
    close(dg - 1, ga.grad, 5e-5, "dgamma (accumulated onto ones)")
    close(db - 1, be.grad, 5e-5, "dbeta (accumulated onto ones)")
    if C % 4 == 0 and (C <= 256 or C % 256 == 0):
        # the launches it replaces: statistics, apply, pool / pool backward, BatchNorm backward
        stat2, rm2, rv2 = torch.empty(2, C, device="cuda"), cu(rm0), cu(rv0)
        L().bn_stats(xd, C, R, C, 1e-5, 0.1, stat2, rm2, rv2, ws())
        close(stat, stat2.cpu(), 1e-6, "stat")
        close(rvd, rv2.cpu(), 1e-6, "running_var vs bn_stats")
        a2 = torch.empty(R, C, device="cuda")
        L().bn_apply(xd, C, stat2, gad, bed, None, C, a2, C, R, C, 1)
        close(act, a2.cpu(), 1e-6, "act vs bn_apply")
        da = dyd
        if kh * kw > 1:
            da = torch.empty(R, C, device="cuda")
            L().maxpool_bwd(act, dyd, da, B, H, W, C, kh, kw)
        dx2, dg2, db2 = torch.empty(R, C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        L().bn_bwd(xd, C, da, C, stat, gad, bed, dx2, C, dg2, db2, R, C, 1, 1, 1, 0, ws())
        close(dx, dx2.cpu(), 2e-5, "dx vs bn_bwd")
        close(dg - 1, dg2.cpu(), 2e-5, "dgamma vs bn_bwd")


@pytest.mark.parametrize("B,chain", [(5, False), (5, True), (64, True), (3, True)])
def test_rpe_fused_steps_vs_torch_gru(B, chain):
    """tatt_rpe_step_fwd / _bwd (one launch per step of the GRU(1024 -> 2x512) that runs over the batch axis with the
    same input at every step) and tatt_rpe_chain_fwd / _bwd (all B steps in one launch of persistent workgroups, B = 64 is the
    BASELINE chain length) against nn.GRU autograd in fp64."""
    torch.manual_seed(0)
    gru = torch.nn.GRU(1024, 512, bidirectional=True).double()
    x = rnd(64, 1024).double()
    out, _ = gru(x.unsqueeze(0).expand(B, 64, 1024))
    dout = rnd(B, 64, 1024, seed=3).double()
    out.backward(dout)
    sfx = ["", "_reverse"]
    W = {n: cu(p_.detach().float()) for n, p_ in gru.named_parameters()}
    gi = torch.stack([x @ getattr(gru, "weight_ih_l0" + d).detach().T + getattr(gru, "bias_ih_l0" + d).detach() for d in sfx])
    gi = cu(gi.float())
    hstep = torch.zeros(2, B + 1, 64, 512, device="cuda")
    gh_all = torch.empty(2, B, 64, 1536, device="cuda")
    qp = torch.empty(B, 16, 64, 64, device="cuda")
    dsh, dsg = (B + 1) * 64 * 512, B * 64 * 1536
    bf, bb = W["bias_hh_l0"], W["bias_hh_l0_reverse"]
    sync = torch.full((L().rpe_chain_sync_words(B),), 7, device="cuda", dtype=torch.int32)      # the call clears its working part itself,
    sync[:8] = 0                                                                                 # the 8-word header (sticky error flags) is the caller's
    if chain:
        assert L().rpe_chain_supported() == 1, "an MI355X in its default partition mode holds the chains (8 XCDs x 32 CUs)"
    if chain:
        L().rpe_chain_fwd(gi, W["weight_hh_l0"], W["weight_hh_l0_reverse"], bf, bb, hstep, gh_all, qp, B, dsg, dsh, sync)
        assert int(sync[1]) == 0, "a reader of the forward chain gave up waiting"
    else:
        for s_ in range(B):
            L().rpe_step_fwd(gi, W["weight_hh_l0"], W["weight_hh_l0_reverse"], bf, bb, hstep[0, s_], hstep[0, s_ + 1], gh_all[0, s_], qp,
                             s_, B, dsg, dsh)
    close(qp.permute(0, 2, 1, 3).reshape(B, 64, 1024), out.detach(), what="rpe out")
    dqp = cu(dout.float().view(B, 64, 16, 64).permute(0, 2, 1, 3))
    carry = torch.zeros(2, 64, 512, device="cuda")
    dgi_sum = torch.zeros(2, 64, 1536, device="cuda")
    dgh_all = torch.empty(2, B, 64, 1536, device="cuda")
    whhT = torch.empty(2, 512, 1536, device="cuda")
    L().rpe_whh_transpose(W["weight_hh_l0"], W["weight_hh_l0_reverse"], whhT)
    assert torch.equal(whhT[1], W["weight_hh_l0_reverse"].T)
    if chain:
        L().rpe_chain_bwd(gi, gh_all, bf, bb, hstep, dqp, carry, dgh_all, dgi_sum, whhT, B, dsg, dsh, sync)
        assert int(sync[1]) == 0, "a reader of the backward chain gave up waiting"
    else:
        for s_ in reversed(range(B)):
            L().rpe_step_bwd(gi, gh_all[0, s_], bf, bb, hstep[0, s_], dqp, carry, dgh_all[0, s_ + 1] if s_ + 1 < B else None,
                             dgh_all[0, s_], dgi_sum, whhT, s_, B, dsg, dsh)
    for d, sx in enumerate(sfx):
        close(dgi_sum[d].sum(0), getattr(gru, "bias_ih_l0" + sx).grad, what="db_ih" + sx)
        close(dgi_sum[d].T @ cu(x.float()), getattr(gru, "weight_ih_l0" + sx).grad, what="dW_ih" + sx)
        close(dgh_all[d].sum((0, 1)), getattr(gru, "bias_hh_l0" + sx).grad, what="db_hh" + sx)
        dwhh = sum(dgh_all[d, s_].T @ hstep[d, s_] for s_ in range(B))
        close(dwhh, getattr(gru, "weight_hh_l0" + sx).grad, what="dW_hh" + sx)


@pytest.mark.parametrize("Lq,p", [(1024, 0.0), (26, 0.0), (1024, 0.1), (100, 0.1), (300, 0.1)])
def test_attention_fwd_bwd(Lq, p):
    B = 3
    Q, K_, V = [rnd(B * n, 64, seed=s).double().requires_grad_(True) for n, s in ((Lq, 1), (26, 2), (26, 3))]
    drop = R.HashDropout(p, 11)
    q = (Q.view(B, Lq, 4, 16).permute(0, 2, 1, 3)) * 0.25
    k = K_.view(B, 26, 4, 16).permute(0, 2, 1, 3)
    v = V.view(B, 26, 4, 16).permute(0, 2, 1, 3)
    att = torch.softmax(q @ k.transpose(-1, -2), -1)
    if p > 0:
        att = att * drop.mask("dec0_attn", tuple(att.shape)).double()
    ctx = (att @ v).permute(0, 2, 1, 3).reshape(B * Lq, 64)
    dctx = rnd(B * Lq, 64, seed=4)
    ctx.backward(dctx.double())
    ss = (11 * 0x85EBCA77 + R.DROP_SITES["dec0_attn"] * 0xC2B2AE3D) & 0xFFFFFFFF
    cfg = (ss, int(p * (1 << 24)), 1 / (1 - p), None) if p > 0 else (0, 0, 1.0, None)
    Qd, Kd, Vd = [cu(t.detach().float()) for t in (Q, K_, V)]
    out, pw = torch.empty(B * Lq, 64, device="cuda"), torch.empty(B, 4, Lq, 26, device="cuda")
    L().attn_fwd(Qd, Kd, Vd, out, pw, B, Lq, 26, *cfg)
    close(out, ctx, what="attn fwd")
    close(pw, att, what="attn weights")
    avg = torch.empty(B, Lq, 26, device="cuda")
    L().attn_avg(pw, avg, B, Lq)
    close(avg, att.mean(1), what="attn avg")
    dQ, dK, dV = torch.empty(B * Lq, 64, device="cuda"), torch.empty(B * 26, 64, device="cuda"), torch.empty(B * 26, 64, device="cuda")
    L().attn_bwd(Qd, Kd, Vd, cu(dctx), dQ, dK, dV, B, Lq, 26, *cfg, 0, ws(L().attn_bwd_workspace(B, Lq)))
    dK2 = dK.clone()
    L().attn_bwd(Qd, Kd, Vd, cu(dctx), dQ, dK2, dV.clone(), B, Lq, 26, *cfg, 1, ws(L().attn_bwd_workspace(B, Lq)))   # accumulate_kv
    close(dK2, 2 * K_.grad, 5e-5, "dK accumulated")
    close(dQ, Q.grad, 5e-5, "dQ")
    close(dK, K_.grad, 5e-5, "dK")
    close(dV, V.grad, 5e-5, "dV")


def test_elementwise_layout_kernels():
    B, C, H, W = 2, 64, 16, 64
    x = rnd(B, C, H, W)
    y = torch.empty(B, H, W, C, device="cuda")
    L().nchw_to_nhwc(cu(x), y, B, C, H * W)
    assert torch.equal(y.cpu(), x.permute(0, 2, 3, 1).contiguous())
    z = torch.empty(B, C, H, W, device="cuda")
    L().nhwc_to_nchw(y, z, B, C, H * W)
    assert torch.equal(z.cpu(), x)
    # prelu fwd/bwd
    xx = rnd(5000).requires_grad_(True)
    a = torch.tensor([0.2], requires_grad=True)
    o = F.prelu(xx, a)
    dy = rnd(5000, seed=1)
    o.backward(dy)
    od = torch.empty(5000, device="cuda")
    L().prelu_fwd(cu(xx.detach()), cu(a.detach()), od, 5000)
    assert torch.equal(od.cpu(), o.detach())
    dxd, da = torch.empty(5000, device="cuda"), torch.zeros(1, device="cuda")
    L().prelu_bwd(cu(xx.detach()), cu(a.detach()), cu(dy), dxd, da, 1, 5000, ws())
    close(dxd, xx.grad, what="prelu dx")
    close(da, a.grad, 1e-4, "prelu dslope")
    # pixel shuffle + mish
    u = rnd(B, 256, H, W).double().requires_grad_(True)
    ps = R.mish(F.pixel_shuffle(u, 2))
    dps = rnd(B, 64, 2 * H, 2 * W, seed=2)
    ps.backward(dps.double())
    un = cu(u.detach().float().permute(0, 2, 3, 1))
    up = torch.empty(B, 2 * H, 2 * W, 64, device="cuda")
    L().pixshuf_mish_fwd(un, up, B, H, W, 64)
    close(up.permute(0, 3, 1, 2), ps, what="pixshuf fwd")
    du = torch.empty(B, H, W, 256, device="cuda")
    L().pixshuf_mish_bwd(un, cu(dps.permute(0, 2, 3, 1)), du, B, H, W, 64)
    close(du.permute(0, 3, 1, 2), u.grad, what="pixshuf bwd")
    # sigmoid + layout
    pre = rnd(B, 2 * H, 2 * W, 4).double().requires_grad_(True)
    sr = torch.sigmoid(pre.permute(0, 3, 1, 2))
    dsr = rnd(B, 4, 2 * H, 2 * W, seed=3)
    sr.backward(dsr.double())
    srd = torch.empty(B, 4, 2 * H, 2 * W, device="cuda")
    L().sigmoid_to_nchw(cu(pre.detach().float()), srd, B, 4, 4 * H * W)
    close(srd, sr, what="sigmoid")
    dpre = torch.empty(B, 2 * H, 2 * W, 4, device="cuda")
    L().sigmoid_bwd_to_nhwc(srd, cu(dsr), dpre, B, 4, 4 * H * W)
    close(dpre, pre.grad, what="sigmoid bwd")
    # max pool
    for kh, kw in ((2, 2), (1, 2)):
        m = rnd(B, 32, 4, 16, seed=kh).requires_grad_(True)
        mp = F.max_pool2d(m, (kh, kw), (kh, kw))
        dmp = rnd(*mp.shape, seed=5)
        mp.backward(dmp)
        mn = cu(m.detach().permute(0, 2, 3, 1))
        mo = torch.empty(B, 4 // kh, 16 // kw, 32, device="cuda")
        L().maxpool_fwd(mn, mo, B, 4, 16, 32, kh, kw)
        assert torch.equal(mo.permute(0, 3, 1, 2).cpu(), mp.detach())
        dm = torch.empty(B, 4, 16, 32, device="cuda")
        L().maxpool_bwd(mn, cu(dmp.permute(0, 2, 3, 1)), dm, B, 4, 16, 32, kh, kw)
        assert torch.equal(dm.permute(0, 3, 1, 2).cpu(), m.grad)
    # axpby, colsum, tp rows, pe
    a_, b_ = rnd(1000), rnd(1000, seed=1)
    o_ = cu(torch.ones(1000))
    L().axpby(cu(a_), 2.0, cu(b_), -0.5, o_, 1000, 1)
    close(o_, 1 + 2 * a_ - 0.5 * b_, what="axpby")
    for R_, C_ in ((3000, 96), (70000, 64), (5, 1536), (9, 40), (4096, 3072), (64, 3072), (130, 2100), (33, 1027), (1, 4)):
        xc = rnd(R_, C_ + 8)
        oc = torch.zeros(C_, device="cuda")
        L().colsum(cu(xc), C_ + 8, R_, C_, oc, 1, ws())
        close(oc, xc[:, :C_].double().sum(0), 1e-5, f"colsum {R_}x{C_}")
    tp = torch.softmax(rnd(26, 3, 37), -1).permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2)     # the reference's non-contiguous view
    rows = torch.empty(3 * 26, 37, device="cuda")
    tpd = tp.cuda()
    assert tpd.stride() == tp.stride()
    L().tp_rows(tpd, tpd.stride(0), tpd.stride(1), tpd.stride(3), rows, 3, 37, 26)
    assert torch.equal(rows.cpu().view(3, 26, 37), tp.permute(0, 3, 1, 2).squeeze(-1))


def test_tps_rotate_loss_adam():
    B, H, W = 3, 16, 64
    bufs = R.tps_buffers()
    x = torch.rand(B, 4, H, W)
    ctrl = (bufs["target_control_points"].unsqueeze(0) + 0.08 * rnd(B, 20, 2)).double().requires_grad_(True)
    P = {"tps." + k: v.double() for k, v in bufs.items()}
    out, _ = R.tps_forward(P, x.double(), ctrl)
    dy = rnd(B, 4, H, W, seed=1)
    out.backward(dy.double())
    T = (bufs["target_coordinate_repr"].double() @ bufs["inverse_kernel"].double()[:, :20]).float()
    y = torch.empty(B, H, W, 4, device="cuda")
    cd = cu(ctrl.detach().float().reshape(B, 40))
    L().tps_sample_fwd(cu(x), cd, cu(T), y, None, B, 4, H, W)
    close(y.permute(0, 3, 1, 2), out, 2e-5, "tps fwd")
    dc = torch.empty(B, 40, device="cuda")
    L().tps_sample_bwd(cu(x), cd, cu(T), cu(dy.permute(0, 2, 3, 1)), dc, B, 4, H, W, 0)
    close(dc.view(B, 20, 2), ctrl.grad, 2e-4, "tps dctrl")
    # rotation (TSC deformation)
    img = torch.rand(B, 4, 32, 128).double().requires_grad_(True)
    arc = torch.tensor([0.1, -0.17, 0.05])
    rot = R.rotate(img, arc.double())
    drot = rnd(B, 4, 32, 128, seed=2)
    rot.backward(drot.double())
    ro = torch.empty(B, 4, 32, 128, device="cuda")
    L().rotate_fwd(cu(img.detach().float()), cu(arc), ro, B, 4, 32, 128)
    assert float((ro.cpu() - rot.detach().float()).abs().max()) < 2e-5
    dimg = torch.zeros(B, 4, 32, 128, device="cuda")
    L().rotate_bwd(cu(arc), cu(drot), dimg, B, 4, 32, 128)
    close(dimg, img.grad, 5e-5, "rotate bwd")
    # a gather with a fixed summation order (no atomics): bit-identical from call to call, and it adds onto what dx holds
    again = torch.zeros(B, 4, 32, 128, device="cuda")
    L().rotate_bwd(cu(arc), cu(drot), again, B, 4, 32, 128)
    assert torch.equal(again, dimg)
    base = cu(rnd(B, 4, 32, 128, seed=9))
    acc = base.clone()
    L().rotate_bwd(cu(arc), cu(drot), acc, B, 4, 32, 128)
    close(acc, base.cpu().double() + img.grad, 5e-5, "rotate bwd (+=)")
    # angles far beyond the +-10 degrees of the training step, three channels, a small odd-sized image: the candidate window of the
    # gather must still hold every output pixel that taps an input pixel
    for (Bq, Cq, Hq, Wq, angles) in ((3, 4, 32, 128, [0.7, -1.3, 3.0]), (2, 3, 7, 10, [0.4, -2.2]), (2, 1, 16, 64, [1.5708, -0.785])):
        im = torch.rand(Bq, Cq, Hq, Wq).double().requires_grad_(True)
        ar = torch.tensor(angles)
        dr = rnd(Bq, Cq, Hq, Wq, seed=4)
        R.rotate(im, ar.double()).backward(dr.double())
        got = torch.zeros(Bq, Cq, Hq, Wq, device="cuda")
        L().rotate_bwd(cu(ar), cu(dr), got, Bq, Cq, Hq, Wq)
        close(got, im.grad, 5e-5, f"rotate bwd {Hq}x{Wq} {angles}")
    # image loss
    hr = torch.rand(B, 4, 32, 128)
    sr = (hr + 0.1 * rnd(B, 4, 32, 128, seed=3)).clamp(0, 1).double().requires_grad_(True)
    per = R.image_loss(sr, hr.double())
    (per.mean() * 100).backward()
    pd = torch.empty(B, device="cuda")
    L().image_loss_fwd(cu(sr.detach().float()), cu(hr), pd, B, 4, 32, 128, 1.0, 1e-4)
    close(pd, per, 1e-5, "image loss")
    dsr = torch.empty(B, 4, 32, 128, device="cuda")
    L().image_loss_bwd(cu(sr.detach().float()), cu(hr), None, 100.0 / B, dsr, B, 4, 32, 128, 1.0, 1e-4)
    close(dsr, sr.grad, 5e-5, "image loss bwd")
    # both in one launch (the training step's form): the same numbers, bit for bit
    pd2, dsr2 = torch.full((B,), float("nan"), device="cuda"), torch.full((B, 4, 32, 128), float("nan"), device="cuda")
    L().image_loss_fwd_bwd(cu(sr.detach().float()), cu(hr), pd2, 100.0 / B, dsr2, B, 4, 32, 128, 1.0, 1e-4)
    assert torch.equal(pd2, pd) and torch.equal(dsr2, dsr)
    # clip + adam (two steps)
    n = 100000
    p0, g1, g2 = rnd(n), rnd(n, seed=1) * 0.01, rnd(n, seed=2) * 0.0001
    pr = p0.clone()
    st = {}
    tot1 = R.clip_and_adam([pr], [g1.clone()], st, 1)
    tot2 = R.clip_and_adam([pr], [g2.clone()], st, 2)
    pd_, m_, v_, state = cu(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(8, device="cuda")
    w = ws()
    L().clip_adam(pd_, cu(g1), m_, v_, n, 1e-3, 0.5, 0.999, 1e-8, 0.25, state, w)
    assert abs(state[0].item() - float(tot1)) < 1e-4 * float(tot1)
    L().clip_adam(pd_, cu(g2), m_, v_, n, 1e-3, 0.5, 0.999, 1e-8, 0.25, state, w)
    assert abs(state[0].item() - float(tot2)) < 1e-4 * float(tot2)
    assert float((pd_.cpu() - pr).abs().max()) < 2e-6


@pytest.mark.parametrize("B,C,H,W,direct", [(3, 4, 32, 128, False), (3, 4, 32, 128, True), (2, 4, 16, 64, False), (1, 3, 9, 13, False),
                                                   (1, 2, 16, 256, False), (1, 1, 4, 1000, False), (2, 1, 64, 64, False)])
def test_tri_ssim_ssim_psnr(B, C, H, W, direct, monkeypatch):
    """TRI_SSIM forward/backward (the TSC loss term) and the SSIM / PSNR metrics against the oracle in fp64.
    The TRI_SSIM launches blur in two 1-D passes with the plane in LDS when the shape fits (32 x 128, 16 x 64, 64 x 64: two strips or
    one; 16 x 256: sixteen strips forward, the direct window walk backward; 4 x 1000: the direct kernels both ways); `direct` forces the
    direct kernels for the training shape as well."""
    if direct:
        monkeypatch.setenv("TATT_SSIM_DIRECT", "1")
    else:
        monkeypatch.delenv("TATT_SSIM_DIRECT", raising=False)
    base = torch.rand(B, C, H, W, generator=torch.Generator().manual_seed(1))
    x1 = (base + 0.05 * rnd(B, C, H, W, seed=2)).clamp(0, 1)
    x2 = (base + 0.05 * rnd(B, C, H, W, seed=3)).clamp(0, 1)
    x3 = base
    a, b_, c = (t.double().requires_grad_(True) for t in (x1, x2, x3))
    ref = R.tri_ssim(a, b_, c)
    ((1 - ref) * 10).backward()
    w = ws(1024)
    out = torch.empty(1, device="cuda")
    L().tri_ssim_fwd(cu(x1), cu(x2), cu(x3), out, B, C, H, W, w)
    assert abs(out.item() - ref.item()) < 2e-6, (out.item(), ref.item())
    base1 = cu(rnd(B, C, H, W, seed=9) * 1e-4)                # accumulate onto something of the gradient's own magnitude
    d1, d2 = base1.clone(), torch.empty(B, C, H, W, device="cuda")
    L().tri_ssim_bwd(cu(x1), cu(x2), cu(x3), -10.0, d1, d2, 1, B, C, H, W)
    L().tri_ssim_bwd(cu(x1), cu(x2), cu(x3), -10.0, d1.clone(), d2, 0, B, C, H, W)
    # sigma = E[x^2] - mu^2 cancels in fp32: the tolerance is what the same fp32 arithmetic on the CPU achieves (x2)
    a32, b32 = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
    ((1 - R.tri_ssim(a32, b32, x3)) * 10).backward()
    tol = max(2e-4, 2 * rel_l2(a32.grad.double().numpy(), a.grad.numpy()))
    close(d1.double() - base1.double(), a.grad, tol, "tri_ssim dx1 (+accumulate)")
    close(d2, b_.grad, tol, "tri_ssim dx2")
    if C >= 3:
        L().ssim_fwd(cu(x1), cu(x3), out, B, C, 3, H, W, w)
        assert abs(out.item() - R.ssim(x1.double(), x3.double()).item()) < 2e-6
        L().psnr(cu(x1), cu(x3), out, B, C, 3, H, W, w)
        assert abs(out.item() - R.psnr(x1.double(), x3.double()).item()) < 1e-4


# ---- GruBlock with the 1x1 conv folded into the GRU input projection -------------------------------------------------------------
@pytest.mark.parametrize("M", [1000, 4096 + 17])
def test_gemm_cat_two_sources_and_two_destinations(M):
    """tatt_gemm_cat: C = [A1 | A2] B^T + bias (forward of a GruBlock on cat(features, text map)) and the split input gradient
    (C, C2) (+)= dY W[:, :64], dY W[:, 64:] in one launch each, against fp64 matmuls."""
    A1, A2, W, b = rnd(M, 64), rnd(M, 64, seed=1), rnd(192, 128, seed=2) * 0.1, rnd(192, seed=3)
    gi = torch.empty(M, 192, device="cuda")
    L().gemm_cat(cu(A1), cu(A2), 64, cu(W), gi, None, 0, cu(b), M, 192, 128, 64, 128, 1, 192, 0, 0, 0, 0)
    ref = torch.cat([A1, A2], 1).double() @ W.double().t() + b.double()
    close(gi, ref, what="cat forward")
    gi1 = torch.empty(M, 192, device="cuda")                                  # single source: plain NT GEMM through the same entry
    L().gemm_cat(cu(A1), None, 0, cu(W[:, :64]), gi1, None, 0, cu(b), M, 192, 64, 64, 64, 1, 192, 0, 0, 0, 0)
    close(gi1, A1.double() @ W[:, :64].double().t() + b.double(), what="single source")
    dY, d1, d2_0 = rnd(M, 192, seed=4), torch.empty(M, 64, device="cuda"), rnd(M, 64, seed=5)
    d2 = cu(d2_0)
    L().gemm_cat(cu(dY), None, 0, cu(W), d1, d2, 64, None, M, 128, 192, 192, 1, 128, 64, 64, 0, 1, 0)
    dref = dY.double() @ W.double()
    close(d1, dref[:, :64], what="dx first half")
    close(d2, d2_0.double() + dref[:, 64:], what="dx second half (accumulated)")
    with pytest.raises(RuntimeError, match="ksplit"):
        L().gemm_cat(cu(A1), cu(A2), 48, cu(W), gi, None, 0, None, M, 192, 128, 64, 128, 1, 192, 0, 0, 0, 0)


@pytest.mark.parametrize("Cin", [64, 128])
def test_gru_fold_and_unfold_vs_autograd(Cin):
    """tatt_gru_fold_w / tatt_gru_unfold_grad against autograd of Wf = cat(W_ih_f, W_ih_r) W_c, bf = cat(W_ih) b_c + cat(b_ih) in fp64."""
    import ctypes
    Wc, bc = rnd(64, Cin, seed=1) * 0.2, rnd(64, seed=2) * 0.1
    Wih = [rnd(96, 64, seed=3 + d) * 0.2 for d in range(2)]
    bih = [rnd(96, seed=5 + d) * 0.1 for d in range(2)]
    dWf, dbf = rnd(192, Cin, seed=7), rnd(192, seed=8)
    leaves = [t.double().requires_grad_(True) for t in (Wc, bc, *Wih, *bih)]
    wc, b_c, wf, wr, bfw, brv = leaves
    Wst = torch.cat([wf, wr], 0)
    Wf_ref, bf_ref = Wst @ wc, Wst @ b_c + torch.cat([bfw, brv])
    ((Wf_ref * dWf.double()).sum() + (bf_ref * dbf.double()).sum()).backward()
    VP = ctypes.c_void_p
    d = {k: cu(v) for k, v in dict(Wc=Wc, bc=bc, wf=Wih[0], wr=Wih[1], bf_=bih[0], br=bih[1], dWf=dWf, dbf=dbf).items()}
    Wf, bf = torch.empty(256, Cin, device="cuda"), torch.empty(192, device="cuda")      # Wf [192][Cin] + WcT [Cin][64]
    p = lambda *ts: [t.data_ptr() for t in ts]
    L().gru_fold_w(1, (VP * 1)(*p(d["Wc"])), (VP * 1)(*p(d["bc"])), (VP * 2)(*p(d["wf"], d["wr"])), (VP * 2)(*p(d["bf_"], d["br"])),
                   (VP * 1)(*p(Wf)), (VP * 1)(*p(bf)), (ctypes.c_int * 1)(Cin))
    close(Wf[:192], Wf_ref, 2e-6, "Wf")
    WcT = Wf.view(-1)[192 * Cin:]
    assert torch.equal(WcT.view(Cin, 64).cpu(), Wc.t())
    close(bf, bf_ref, 2e-6, "bf")
    g0 = [rnd(*t.shape, seed=20 + i) for i, t in enumerate((Wih[0], Wih[1], bih[0], bih[1], Wc, bc))]       # += onto existing gradients
    g = [cu(t) for t in g0]
    L().gru_unfold_grad(d["dWf"], d["dbf"], WcT, d["bc"], d["wf"], d["wr"], g[0], g[1], g[2], g[3], g[4], g[5], Cin)
    for got, base, leaf, nm in zip(g, g0, (wf, wr, bfw, brv, wc, b_c), ("dW_ih_f", "dW_ih_r", "db_ih_f", "db_ih_r", "dW_c", "db_c")):
        close(got, base.double() + leaf.grad, 2e-6, nm)


# ---- BatchNorm fused into conv / GEMM epilogues ------------------------------------------------------------------------------------
def _mish(t):
    return t * torch.tanh(F.softplus(t))


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(3, 16, 64, 64, 64), (2, 8, 24, 32, 64), (1, 5, 13, 64, 256)])
def test_conv_fwd_bn_statistics_in_the_epilogue(B, H, W, Cin, Cout):
    """tatt_conv_fwd_bn(epi 1) + tatt_bn_stats_from_partial = conv followed by train-mode BatchNorm statistics (ragged last row block)."""
    x, w, b = rnd(B, Cin, H, W), rnd(Cout, Cin, 3, 3, seed=1) / math.sqrt(9 * Cin), rnd(Cout, seed=2)
    y = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    R_ = B * H * W
    wf, wd = torch.empty(w.numel(), device="cuda"), torch.empty(w.numel(), device="cuda")
    L().conv_pack_w(cu(w), wf, wd, Cout, Cin, 3, 3)
    out = torch.empty(B, H, W, Cout, device="cuda")
    nblk = -(-R_ // 128)
    part = torch.full((nblk * 2 * Cout,), float("nan"), device="cuda")
    L().conv_fwd_bn(cu(x.permute(0, 2, 3, 1)), wf, cu(b), out, B, H, W, Cin, Cout, 3, 3, Cin, Cout, 9 * Cin, 1, 0, Cin, 0, 1,
                    None, 0, None, None, None, 0, part)
    close(out.permute(0, 3, 1, 2), y, what="conv")
    stat, rm, rv = torch.empty(2, Cout, device="cuda"), cu(rnd(Cout, seed=3)), cu(rnd(Cout, seed=4).abs() + 0.5)
    rm0, rv0 = rm.clone(), rv.clone()
    L().bn_stats_from_partial(part, nblk, 128, R_, Cout, 1e-5, 0.1, stat, rm, rv)
    mean, var = y.mean((0, 2, 3)), y.var((0, 2, 3), unbiased=False)
    close(stat[0], mean, 2e-5, "mean")
    close(stat[1], 1 / torch.sqrt(var + 1e-5), 2e-5, "invstd")
    close(rm, 0.9 * rm0.cpu().double() + 0.1 * mean, 2e-5, "running_mean")
    close(rv, 0.9 * rv0.cpu().double() + 0.1 * y.var((0, 2, 3), unbiased=True), 2e-5, "running_var")


@pytest.mark.parametrize("C,out16", [(64, False), (64, True), (256, False)])
def test_grouped_batchnorm_final_and_apply_equal_one_launch_per_group(C, out16):
    """The batched TSC step's BatchNorm: tatt_bn_stats_from_partial_groups / tatt_bn_apply(_bf16)_groups over two passes in ONE launch each
    against the per-pass launches, bit for bit -- statistics, running statistics (updated pass after pass) and the normalised rows."""
    Rg, ng, act = 2048, 2, 2
    R_, nblk = Rg * ng, Rg // 128
    x = cu(rnd(R_, C) * 1.3 + 0.2)
    x[Rg:] *= 0.5                                                            # the passes have different statistics
    part = torch.empty(ng, nblk, 2, C, device="cuda")
    xb = x.view(ng, nblk, 128, C)
    part[:, :, 0] = xb.mean(2)
    part[:, :, 1] = ((xb - xb.mean(2, keepdim=True)) ** 2).sum(2)
    ga, be, res = cu(1 + 0.1 * rnd(C, seed=1)), cu(0.1 * rnd(C, seed=2)), cu(rnd(R_, C, seed=3))
    rm0, rv0 = cu(rnd(C, seed=4)), cu(rnd(C, seed=5).abs() + 0.5)
    ydt = torch.bfloat16 if out16 else torch.float32
    # per pass
    s1, rm1, rv1 = torch.empty(ng, 2, C, device="cuda"), rm0.clone(), rv0.clone()
    y1 = torch.empty(R_, C, device="cuda", dtype=ydt)
    for g in range(ng):
        L().bn_stats_from_partial(part[g], nblk, 128, Rg, C, 1e-5, 0.1, s1[g], rm1, rv1)
        (L().bn_apply_bf16 if out16 else L().bn_apply)(x[g * Rg:], C, s1[g], ga, be, res[g * Rg:], C, y1[g * Rg:], C, Rg, C, act)
    # grouped
    s2, rm2, rv2 = torch.empty(ng, 2, C, device="cuda"), rm0.clone(), rv0.clone()
    y2 = torch.empty(R_, C, device="cuda", dtype=ydt)
    L().bn_stats_from_partial_groups(part, nblk, 128, Rg, C, ng, 1e-5, 0.1, s2, rm2, rv2)
    (L().bn_apply_bf16_groups if out16 else L().bn_apply_groups)(x, C, s2, ga, be, res, C, y2, C, R_, C, act, Rg)
    assert torch.equal(s1, s2) and torch.equal(rm1, rm2) and torch.equal(rv1, rv2) and torch.equal(y1, y2)
    assert not torch.equal(s1[0], s1[1])
    with pytest.raises(RuntimeError):
        L().bn_apply_groups(x, C, s2, ga, be, res, C, y2.float(), C, R_, C, act, Rg - 1)        # rows per group must fit the blocks


@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(3, 16, 64, 64, 64), (1, 7, 19, 64, 256)])
def test_conv_dgrad_with_batchnorm_backward_in_the_epilogue(B, H, W, Cin, Cout, act):
    """tatt_conv_fwd_bn(epi 2) + tatt_bn_bwd_from_partial against autograd of z = conv(act(BatchNorm_train(x))) in fp64."""
    x0, w, dz = rnd(B, Cin, H, W) * 1.5 + 0.3, rnd(Cout, Cin, 3, 3, seed=1) / math.sqrt(9 * Cin), rnd(B, Cout, H, W, seed=2)
    ga0, be0 = 1 + 0.2 * rnd(Cin, seed=3), 0.1 * rnd(Cin, seed=4)
    x, ga, be = (t.double().requires_grad_(True) for t in (x0, ga0, be0))
    bn = F.batch_norm(x, None, None, ga, be, True, 0.1, 1e-5)
    a = bn if act == 0 else (F.relu(bn) if act == 1 else _mish(bn))
    F.conv2d(a, w.double(), None, padding=1).backward(dz.double())
    R_ = B * H * W
    xn = cu(x0.permute(0, 2, 3, 1))
    stat, ws_ = torch.empty(2, Cin, device="cuda"), ws()
    L().bn_stats(xn, Cin, R_, Cin, 1e-5, 0.1, stat, None, None, ws_)
    wf, wd = torch.empty(w.numel(), device="cuda"), torch.empty(w.numel(), device="cuda")
    L().conv_pack_w(cu(w), wf, wd, Cout, Cin, 3, 3)
    nblk = -(-R_ // 128)
    g, part = torch.empty(R_, Cin, device="cuda"), torch.full((nblk * 2 * Cin,), float("nan"), device="cuda")
    L().conv_fwd_bn(cu(dz.permute(0, 2, 3, 1)), wd, None, g, B, H, W, Cout, Cin, 3, 3, Cout, Cin, 9 * Cout, 1, 0, Cout, 0, 2,
                    xn, Cin, stat, cu(ga0), cu(be0), act, part)
    dx, dga, dbe = torch.empty(R_, Cin, device="cuda"), cu(rnd(Cin, seed=5)), cu(rnd(Cin, seed=6))
    dga0, dbe0 = dga.clone(), dbe.clone()
    L().bn_bwd_from_partial(xn, Cin, g, Cin, stat, cu(ga0), cu(be0), part, nblk, dx, Cin, dga, dbe, R_, Cin, 1, 1, 0, torch.empty(2 * Cin, device="cuda"))
    close(dx.view(B, H, W, Cin).permute(0, 3, 1, 2), x.grad, 5e-5, "dx")
    close(dga - dga0, ga.grad, 5e-5, "dgamma")
    close(dbe - dbe0, be.grad, 5e-5, "dbeta")


def test_gemm_cat_bnbwd_matches_separate_batchnorm_backward():
    """tatt_gemm_cat_bnbwd: the split input gradient of a GruBlock whose first input is a BatchNorm output -- first destination reduced
    for that BatchNorm in the epilogue, second destination accumulated as usual -- against autograd in fp64."""
    M = 3 * 16 * 64 + 40
    x0, tp0, Wf, dgi = rnd(M, 64) * 1.3 + 0.2, rnd(M, 64, seed=1), rnd(192, 128, seed=2) * 0.1, rnd(M, 192, seed=3)
    ga0, be0, d2_0 = 1 + 0.2 * rnd(64, seed=4), 0.1 * rnd(64, seed=5), rnd(M, 64, seed=6)
    x, ga, be, tp = (t.double().requires_grad_(True) for t in (x0, ga0, be0, tp0))
    bn = F.batch_norm(x, None, None, ga, be, True, 0.1, 1e-5)
    (torch.cat([bn, tp], 1) @ Wf.double().t()).backward(dgi.double())
    xg = cu(x0)
    stat = torch.empty(2, 64, device="cuda")
    L().bn_stats(xg, 64, M, 64, 1e-5, 0.1, stat, None, None, ws())
    nblk = -(-M // 128)
    g, d2, part = torch.empty(M, 64, device="cuda"), cu(d2_0), torch.full((nblk * 128,), float("nan"), device="cuda")
    L().gemm_cat_bnbwd(cu(dgi), None, 0, cu(Wf), g, d2, 64, None, M, 128, 192, 192, 1, 128, 64, 64, 0, 1, 0, xg, 64, stat, cu(ga0), cu(be0),
                       0, 64, part)
    close(d2, d2_0.double() + tp.grad, what="second destination")
    dx, dga, dbe = torch.empty(M, 64, device="cuda"), torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda")
    L().bn_bwd_from_partial(xg, 64, g, 64, stat, cu(ga0), cu(be0), part, nblk, dx, 64, dga, dbe, M, 64, 1, 1, 0, torch.empty(128, device="cuda"))
    close(dx, x.grad, 5e-5, "dx")
    close(dga, ga.grad, 5e-5, "dgamma")
    close(dbe, be.grad, 5e-5, "dbeta")

