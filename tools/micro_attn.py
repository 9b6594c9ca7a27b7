"""Standalone timing of the decoder-shaped attention kernels (B=64, Lq=1024, 26 keys): python tools/micro_attn.py"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tatt_amd.lib import lib
L = lib()
B, Lq = 64, 1024
Q = torch.randn(B * Lq, 64, device="cuda"); K = torch.randn(B * 26, 64, device="cuda"); V = torch.randn(B * 26, 64, device="cuda")
ctx = torch.empty_like(Q); dctx = torch.randn_like(Q); dQ = torch.empty_like(Q); dK = torch.empty_like(K); dV = torch.empty_like(V)
ws = torch.empty(L.attn_bwd_workspace(B, Lq), device="cuda")
cfg = (12345, int(0.1 * (1 << 24)), 1 / 0.9, None)
for name, fn in {"attn_fwd": lambda: L.attn_fwd(Q, K, V, ctx, None, B, Lq, 26, *cfg),
                 "attn_bwd": lambda: L.attn_bwd(Q, K, V, dctx, dQ, dK, dV, B, Lq, 26, *cfg, 0, ws)}.items():
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); e1.synchronize()
    print(f"{name:10s} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
