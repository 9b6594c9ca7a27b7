"""A/B timing of two builds of the library in one process: python tools/ab_conv.py libA.so libB.so
Alternates the two builds so that clock / box differences cancel."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tatt_amd.lib as TL
B = 64; R = B * 1024
libs = []
for pth in sys.argv[1:3]:
    TL.LIB_PATH = os.path.abspath(pth); TL._lib = None
    libs.append(TL.lib())
rnd = lambda *s: torch.randn(*s, device="cuda")
x, y, w, b = rnd(R, 64), rnd(R, 64), rnd(64, 64, 3, 3), rnd(64)
wf, wd, dw, ws = rnd(64 * 576), rnd(64 * 576), rnd(64, 64, 3, 3), torch.empty(48 << 20, device="cuda")
x256, w256 = rnd(R, 256), rnd(256 * 576)
cases = {
    "conv fwd packed 64->64": lambda L: L.conv_fwd(x, wf, b, y, B, 16, 64, 64, 64, 3, 3, 64, 64, 576, 1, 0, 64, 0, 0),
    "conv fwd packed 64->256": lambda L: L.conv_fwd(x, w256, None, x256, B, 16, 64, 64, 256, 3, 3, 64, 256, 576, 1, 0, 64, 0, 0),
    "wgrad 64->64 sk128": lambda L: L.conv_wgrad(x, y, dw, B, 16, 64, 64, 64, 3, 3, 64, 64, 128, ws, 1),
    "gemm NT K=64": lambda L: L.gemm(x, w, y, b, R, 64, 64, 64, 1, 64, 1, 64, 1, 0, 0, 0, 0, 0, 1, None),
}
def t(fn, L, iters=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn(L)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, fn in cases.items():
    for L in libs: t(fn, L, 5)
    res = [[], []]
    for rep in range(5):
        for i, L in enumerate(libs): res[i].append(t(fn, L))
    print(f"{name:28s} A {min(res[0]):7.1f} us (med {sorted(res[0])[2]:7.1f})   B {min(res[1]):7.1f} us (med {sorted(res[1])[2]:7.1f})")
