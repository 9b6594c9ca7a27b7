"""TATT_F32X3 (three bf16 pieces per operand, six MFMA terms) against the fp32 MFMA path and an fp64 reference: error and time of the
SRB-shaped 3x3 convolution and of a [65536,128]x[128,192] GEMM.  python tools/micro_x3.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tatt_amd.lib import lib
L = lib()
torch.manual_seed(0)
B, H, W, Cin, Cout = 64, 16, 64, 64, 64
x = torch.randn(B, H, W, Cin, device="cuda")
w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
bias = torch.randn(Cout, device="cuda")
wf = w.permute(0, 2, 3, 1).contiguous().view(-1)      # [n][tap][ci]
ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
ws = torch.empty(1 << 20, device="cuda")


def run(fl, iters=50):
    y = torch.zeros(B, H, W, Cout, device="cuda")
    args = (x, wf, bias, y, B, H, W, Cin, Cout, 3, 3, Cin, Cout, 9 * Cin, 1, 0, Cin, 0, fl, 1, ws)
    L.conv_fwd_splitk(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        L.conv_fwd_splitk(*args)
    e1.record()
    e1.synchronize()
    err = (y.double() - ref)
    return float(err.abs().max()), float(err.norm() / ref.norm()), e0.elapsed_time(e1) / iters * 1e3


for name, fl in (("fp32 MFMA", 0), ("f32x3", 4), ("bf16", 2)):
    mx, rel, us = run(fl)
    print(f"conv3x3 64->64 [64,16,64] {name:10s} max-abs {mx:.3e} rel-L2 {rel:.3e}  {us:6.1f} us")

M, K, N = 65536, 128, 192
a = torch.randn(M, K, device="cuda")
bw = torch.randn(N, K, device="cuda") * 0.1
refg = a.double() @ bw.double().t()
for name, fl in (("fp32 MFMA", 0), ("f32x3", 4)):
    c = torch.zeros(M, N, device="cuda")
    args = (a, bw, c, None, M, N, K, K, 1, K, 1, N, 1, 0, 0, 0, 0, fl, 1, None)
    L.gemm(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        L.gemm(*args)
    e1.record()
    e1.synchronize()
    err = c.double() - refg
    print(f"gemm [65536,128]x[128,192]  {name:10s} max-abs {float(err.abs().max()):.3e} rel-L2 {float(err.norm() / refg.norm()):.3e}  {e0.elapsed_time(e1) / 50 * 1e3:6.1f} us")

# weight gradient of the SRB convolution (row-contiguous operands: im2col(X)^T on M, dY on N) and a dX product (B row-contiguous)
dy = torch.randn(B, H, W, Cout, device="cuda")
wsp = torch.empty(16 << 20, device="cuda")
gw = None
for name, fl in (("fp32 MFMA", 0), ("f32x3", 4)):
    dW = torch.zeros(Cout, Cin, 3, 3, device="cuda")
    args = (x, dy, dW, B, H, W, Cin, Cout, 3, 3, Cin, Cout, 154, wsp, fl)
    L.conv_wgrad(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        L.conv_wgrad(*args)
    e1.record()
    e1.synchronize()
    if gw is None:
        gw = dW.double()
    print(f"conv wgrad 64->64 (+ slab reduce) {name:10s} rel-L2 vs fp32 path {float((dW.double() - gw).norm() / gw.norm()):.3e}  {e0.elapsed_time(e1) / 50 * 1e3:6.1f} us")
g = torch.randn(M, N, device="cuda")
refd = g.double() @ bw.double()
for name, fl in (("fp32 MFMA", 0), ("f32x3", 4)):
    dx = torch.zeros(M, K, device="cuda")
    args = (g, bw, dx, None, M, K, N, N, 1, 1, K, K, 1, 0, 0, 0, 0, fl, 1, None)
    L.gemm(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        L.gemm(*args)
    e1.record()
    e1.synchronize()
    err = dx.double() - refd
    print(f"gemm dX [65536,192]x[192,128] {name:10s} rel-L2 {float(err.norm() / refd.norm()):.3e}  {e0.elapsed_time(e1) / 50 * 1e3:6.1f} us")
