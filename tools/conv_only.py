"""Runs only the SRB conv (fwd, dgrad, wgrad) a few times: target for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tatt_amd.lib import lib
L = lib(); B = 64; R = B * 1024
x, y, w, b, dw = [torch.randn(*s, device="cuda") for s in ((R, 64), (R, 64), (64, 64, 3, 3), (64,), (64, 64, 3, 3))]
ws = torch.empty(48 << 20, device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    L.conv_fwd(x, w, b, y, B, 16, 64, 64, 64, 3, 3, 64, 64, 576, 9, 0, 1, 0, 0)
    L.conv_wgrad(x, y, dw, B, 16, 64, 64, 64, 3, 3, 64, 64, 205, ws, 1)
torch.cuda.synchronize()
