"""Standalone timings (B=64 SRB shapes) of the BatchNorm-epilogue launches and their second halves: python tools/micro_bn.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tatt_amd.lib import lib
L = lib()
B, H, W, C = 64, 16, 64, 64
R = B * H * W
dev = "cuda"
x = torch.randn(R, C, device=dev); y = torch.empty(R, C, device=dev); g = torch.randn(R, C, device=dev)
w = torch.randn(C * 9 * C, device=dev) * 0.05; b = torch.randn(C, device=dev)
stat = torch.rand(2, C, device=dev) + 0.5; ga = torch.rand(C, device=dev) + 0.5; be = torch.randn(C, device=dev) * 0.1
nblk = R // 128
part = torch.empty(nblk * 2 * C, device=dev); sums = torch.empty(2 * C, device=dev)
dga, dbe = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
dgi = torch.randn(R, 192, device=dev); Wf = torch.randn(192, 128, device=dev) * 0.1; d1 = torch.empty(R, C, device=dev); d2 = torch.zeros(R, C, device=dev)
ws = torch.empty(1 << 20, device=dev)
rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
conv = (x, w, b, y, B, H, W, C, C, 3, 3, C, C, 9 * C, 1, 0, C)
tests = {
    "conv fwd plain": lambda: L.conv_fwd(*conv, 0, 0),
    "conv fwd epi1 (stats)": lambda: L.conv_fwd_bn(*conv, 0, 1, None, 0, None, None, None, 0, part),
    "conv dgrad epi2 act0": lambda: L.conv_fwd_bn(*conv[:2], None, *conv[3:], 0, 2, g, C, stat, ga, be, 0, part),
    "conv dgrad epi2 mish": lambda: L.conv_fwd_bn(*conv[:2], None, *conv[3:], 0, 2, g, C, stat, ga, be, 2, part),
    "gemm_cat dx plain": lambda: L.gemm_cat(dgi, None, 0, Wf, d1, d2, 64, None, R, 128, 192, 192, 1, 128, 64, 64, 0, 1, 0),
    "gemm_cat dx bnbwd": lambda: L.gemm_cat_bnbwd(dgi, None, 0, Wf, d1, d2, 64, None, R, 128, 192, 192, 1, 128, 64, 64, 0, 1, 0, x, C, stat, ga, be, 0, 64, part),
    "bn_stats_from_partial": lambda: L.bn_stats_from_partial(part, nblk, 128, R, C, 1e-5, 0.1, stat, rm, rv),
    "bn_apply mish": lambda: L.bn_apply(x, C, stat, ga, be, None, C, y, C, R, C, 2),
    "bn_bwd_from_partial": lambda: L.bn_bwd_from_partial(x, C, g, C, stat, ga, be, part, nblk, y, C, dga, dbe, R, C, 1, 1, 0, sums),
    "bn_bwd (3 kernels) mish": lambda: L.bn_bwd(x, C, g, C, stat, ga, be, y, C, dga, dbe, R, C, 2, 1, 1, 0, ws),
    "bn_stats (2 kernels)": lambda: L.bn_stats(x, C, R, C, 1e-5, 0.1, stat, rm, rv, ws),
}
for name, fn in tests.items():
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); e1.synchronize()
    stat.copy_(torch.rand(2, C, device=dev) + 0.5)
    print(f"{name:28s} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
