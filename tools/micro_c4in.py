import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tatt_amd.lib import lib
L = lib(); B = 64
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ws = torch.empty(8 << 20, device="cuda")
for (H, W, Cin, Cout, k) in ((16, 64, 4, 32, 3), (16, 64, 4, 64, 9), (8, 32, 32, 64, 3)):
    R = B * H * W; T = k * k
    x = torch.randn(R, Cin, device="cuda"); w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.1
    wf, wd = torch.empty(w.numel(), device="cuda"), torch.empty(w.numel(), device="cuda")
    L.conv_pack_w(w, wf, wd, Cout, Cin, k, k)
    b = torch.randn(Cout, device="cuda"); y = torch.empty(R, Cout, device="cuda")
    us = t(lambda: L.conv_fwd_splitk(x, wf, b, y, B, H, W, Cin, Cout, k, k, Cin, Cout, T * Cin, 1, 0, Cin, 0, 0, 1, ws))
    print(f"conv {k}x{k} {Cin}->{Cout} on [{B},{H},{W}]: {us:.1f} us  ({2.0 * R * Cout * T * Cin / us / 1e6:.1f} TF)")
