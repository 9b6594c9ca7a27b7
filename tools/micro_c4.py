"""Kernel durations of the direct 9x9 Cout = 4 convolution (the 64 -> 4 output conv on 32x128 maps, and block1's input gradient on
16x64 maps).  bash tools/run_micro.sh tools/micro_c4.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tatt_amd.lib import lib
L = lib()
B = 64
for (H, W) in [(32, 128), (16, 64)]:
    R = B * H * W
    x, w, b = torch.randn(R, 64, device="cuda"), torch.randn(4, 64, 9, 9, device="cuda") * 0.02, torch.randn(4, device="cuda")
    y = torch.empty(R, 4, device="cuda")
    z = torch.empty(L.conv_skinny_workspace(B, H, W, 64, 4, 9, 9), device="cuda")
    for _ in range(6):
        L.conv_skinny_fwd(x, w, b, y, B, H, W, 64, 4, 9, 9, 64, 4, 64 * 81, 81, 0, 1, 0, z)
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x.view(B, H, W, 64).permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=4).permute(0, 2, 3, 1).reshape(R, 4)
    print(H, W, "rel err", float((y.double() - ref).norm() / ref.norm()))
