"""gru32 fwd / bwd in both orientations at B=64 under rocprofv3 --kernel-trace: bash tools/run_micro.sh tools/micro_gru.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tatt_amd.lib import lib
L = lib()
B, H, W = 64, 16, 64
R = B * H * W
g = lambda *s: torch.randn(*s, device="cuda") * 0.3
gi, whh, bhh, out, sv = g(R, 192), g(96, 32), g(96), torch.empty(R, 64, device="cuda"), torch.empty(R, 2, 4, 32, device="cuda")
dout, dgi, dghn, hp = g(R, 64), torch.empty(R, 192, device="cuda"), torch.empty(R, 64, device="cuda"), torch.empty(R, 64, device="cuda")
for name, sm in (("horizontal", (B * H, W, 1, W, 0, 1)), ("vertical", (B * W, H, W, H * W, 1, W))):
    for _ in range(5):
        L.gru32_fwd(gi, whh, whh, bhh, bhh, out, sv, *sm)
        L.gru32_bwd(dout, out, sv, whh, whh, dgi, dghn, hp, *sm)
    torch.cuda.synchronize()
