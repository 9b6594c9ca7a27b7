"""[R,64] x [64,64]^T linears at growing R: does a second / fourth round of workgroups overlap with the first? (bash tools/run_micro.sh tools/micro_lin.py)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tatt_amd.lib import lib
L = lib()
W = torch.randn(64, 64, device="cuda")
for R in (16384, 32768, 65536, 131072, 262144):
    x, y = torch.randn(R, 64, device="cuda"), torch.empty(R, 64, device="cuda")
    for _ in range(6):
        L.gemm(x, W, y, None, R, 64, 64, 64, 1, 64, 1, 64, 1, 0, 0, 0, 0, 0, 1, None)
    torch.cuda.synchronize()
