"""Per-shape timing of bench.py's conv launch mix (3x3 conv fwd / dgrad as launched by the engine, split-K reduce included where the
engine splits): python tools/conv_mix.py [B]"""
import sys
import torch
sys.path.insert(0, ".")
from tatt_amd.lib import lib
from tatt_amd.engine import Engine

L = lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
shapes = [(16, 64, 64, 64), (16, 64, 64, 256), (16, 64, 256, 64), (16, 32, 32, 64), (16, 32, 64, 32), (8, 16, 64, 128), (8, 16, 128, 64),
          (4, 8, 128, 256), (4, 8, 256, 128), (2, 4, 256, 256), (2, 2, 256, 256)]
ws = torch.empty(48 << 20, device="cuda")
for H, W, Cin, Cout in shapes:
    x = torch.randn(B, H, W, Cin, device="cuda")
    wf = torch.randn(Cout * 9 * Cin, device="cuda") * 0.05
    b = torch.randn(Cout, device="cuda")
    y = torch.empty(B, H, W, Cout, device="cuda")
    sk = Engine._conv_splitk(B * H * W, Cout, 9 * Cin)
    args = (x, wf, b, y, B, H, W, Cin, Cout, 3, 3, Cin, Cout, 9 * Cin, 1, 0, Cin, 0, 0, sk, ws)
    for _ in range(3):
        L.conv_fwd_splitk(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.conv_fwd_splitk(*args)
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * B * H * W * Cout * 9 * Cin
    print(f"{H:3d}x{W:3d} {Cin:4d}->{Cout:4d} splitk {sk:3d}  {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s  ({fl / 1e9:.2f} GF)")
