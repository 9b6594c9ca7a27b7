"""Runs the fused RPE steps alone (target for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tatt_amd.lib import lib
L = lib(); B = 64
rnd = lambda *s: torch.randn(*s, device="cuda")
gi_, whf, whb, bhf, bhb = rnd(2, 64, 1536), rnd(1536, 512) * 0.05, rnd(1536, 512) * 0.05, rnd(1536), rnd(1536)
hst, gha, qp_ = torch.zeros(2, B + 1, 64, 512, device="cuda"), rnd(2, B, 64, 1536), rnd(B, 16, 64, 64)
dsh_, dsg_ = (B + 1) * 64 * 512, B * 64 * 1536
whT, car, dga, dgs = rnd(2, 512, 1536), rnd(2, 64, 512), rnd(2, B, 64, 1536), torch.zeros(2, 64, 1536, device="cuda")
for it in range(3):
    for s in range(B):
        L.rpe_step_fwd(gi_, whf, whb, bhf, bhb, hst[0, s], hst[0, s + 1], gha[0, s], qp_, s, B, dsg_, dsh_)
    for s in reversed(range(B)):
        L.rpe_step_bwd(gi_, gha[0, s], bhf, bhb, hst[0, s], qp_, car, dga[0, s + 1] if s + 1 < B else None, dga[0, s], dgs, whT, s, B, dsg_, dsh_)
torch.cuda.synchronize()
