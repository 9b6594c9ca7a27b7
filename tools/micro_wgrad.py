"""3x3 64->64 conv weight gradient at B=64 for a range of split-K factors (kernel durations: run under rocprofv3 --kernel-trace,
bash tools/run_micro.sh tools/micro_wgrad.py; the grid's z extent in the trace is the split factor)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tatt_amd.lib import lib
L = lib()
B, H, W, C = 64, 16, 64, 64
R = B * H * W
x, dy = torch.randn(R, C, device="cuda"), torch.randn(R, C, device="cuda")
dW = torch.zeros(C, C, 3, 3, device="cuda")
ws = torch.empty(64 << 20, device="cuda")
sks = [int(a) for a in sys.argv[1:]] or [51, 64, 77, 85, 96, 102, 128, 154, 170, 204, 256]
nsk = len(sks)
for sk in sks:
    for _ in range(6):
        L.conv_wgrad(x, dy, dW, B, H, W, C, C, 3, 3, C, C, sk, ws, 1)
torch.cuda.synchronize()
ref = torch.nn.grad.conv2d_weight(x.view(B, H, W, C).permute(0, 3, 1, 2), dW.shape, dy.view(B, H, W, C).permute(0, 3, 1, 2), padding=1)
print("rel err", float((dW / (6 * nsk) - ref).norm() / ref.norm()))
