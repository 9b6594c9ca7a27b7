"""Kernel durations of the STN head's small-map BatchNorm + ReLU + max-pool: the fused launch (tatt_bn_pool_small_*) beside the
generic launches it replaces.  bash tools/run_micro.sh tools/micro_stn_bn.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tatt_amd.lib import lib
L = lib()
B = 64
ws = torch.empty(1 << 22, device="cuda")
for (H, W, C, kh, kw) in [(4, 16, 128, 2, 2), (2, 8, 256, 2, 2), (1, 4, 256, 1, 2), (1, 2, 256, 1, 1), (1, 1, 512, 1, 1)]:
    R, Rp = B * H * W, B * (H // kh) * (W // kw)
    x, dy = torch.randn(R, C, device="cuda"), torch.randn(Rp, C, device="cuda")
    ga, be = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    stat, act, pooled = torch.empty(2, C, device="cuda"), torch.empty(R, C, device="cuda"), torch.empty(Rp, C, device="cuda")
    dx, da, dg, db = torch.empty(R, C, device="cuda"), torch.empty(R, C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    for _ in range(5):
        L.bn_pool_small_fwd(x, stat, ga, be, rm, rv, act, pooled, B, H, W, C, kh, kw, 1e-5, 0.1)
        L.bn_pool_small_bwd(x, act, dy, stat, ga, dx, dg, db, B, H, W, C, kh, kw)
        L.bn_stats(x, C, R, C, 1e-5, 0.1, stat, rm, rv, ws)
        L.bn_apply(x, C, stat, ga, be, None, C, act, C, R, C, 1)
        if kh * kw > 1:
            L.maxpool_fwd(act, pooled, B, H, W, C, kh, kw)
            L.maxpool_bwd(act, dy, da, B, H, W, C, kh, kw)
        L.bn_bwd(x, C, da, C, stat, ga, be, dx, C, dg, db, R, C, 1, 1, 1, 0, ws)
torch.cuda.synchronize()
