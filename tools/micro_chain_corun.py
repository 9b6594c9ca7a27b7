"""How much does a persistent chain launch (tatt_rpe_chain_bwd, 256 workgroups x 512 threads polling flags) slow down OTHER work that runs
beside it?  A dependent chain of 60 small launches, the block1 9x9 dgrad and a big bandwidth-bound launch, alone and beside the chain,
eagerly and as a captured graph.  python tools/micro_chain_corun.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tatt_amd.lib import lib
L = lib()
B = 64
g = lambda *s: torch.randn(*s, device="cuda")
gi = g(2, 64, 1536) * 0.5
whf, whb, bhf, bhb = g(1536, 512) * 0.04, g(1536, 512) * 0.04, g(1536) * 0.1, g(1536) * 0.1
dsh, dsg = (B + 1) * 64 * 512, B * 64 * 1536
hst, gha, qp = torch.zeros(2, B + 1, 64, 512, device="cuda"), torch.empty(2, B, 64, 1536, device="cuda"), torch.empty(B, 16, 64, 64, device="cuda")
sync = torch.zeros(L.rpe_chain_sync_words(B), device="cuda", dtype=torch.int32)
L.rpe_chain_fwd(gi, whf, whb, bhf, bhb, hst, gha, qp, B, dsg, dsh, sync)
whT = torch.empty(2, 512, 1536, device="cuda")
L.rpe_whh_transpose(whf, whb, whT)
dqp = g(B, 16, 64, 64)
sync2 = torch.zeros(L.rpe_chain_sync_words(B), device="cuda", dtype=torch.int32)
car, dga, dgs = torch.zeros(2, 64, 512, device="cuda"), torch.empty(2, B, 64, 1536, device="cuda"), torch.zeros(2, 64, 1536, device="cuda")


def chain():
    car.zero_(); dgs.zero_()
    L.rpe_chain_bwd(gi, gha, bhf, bhb, hst, dqp, car, dga, dgs, whT, B, dsg, dsh, sync2)


small = [g(64, 512) for _ in range(2)]
x9 = g(B * 16 * 64, 64)
w9 = g(64, 4, 9, 9) * 0.05
dxs = torch.empty(B * 16 * 64, 4, device="cuda")
z9 = torch.empty(L.conv_skinny_workspace(B, 16, 64, 64, 4, 9, 9), device="cuda")
big = [g(B * 1024, 64) for _ in range(3)]


def small_chain():
    for i in range(60):
        L.axpby(small[i & 1], 1.0, None, 0.0, small[1 - (i & 1)], 64 * 512, 0)


def c4():
    L.conv_skinny_fwd(x9, w9, None, dxs, B, 16, 64, 64, 4, 9, 9, 64, 4, 81, 4 * 81, 80, -1, 0, z9)


def bigk():
    for _ in range(4):
        L.axpby(big[0], 1.0, big[1], 1.0, big[2], B * 1024 * 64, 0)


side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def timed(fn, beside, graph):
    def body():
        if beside:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                chain()
        fn()
        if beside:
            torch.cuda.current_stream().wait_stream(side)
    if graph:
        gr = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            body()
            torch.cuda.synchronize()
            with torch.cuda.graph(gr, stream=s):
                body()
        run = gr.replay
    else:
        run = body
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3


for graph in (False, True):
    t_chain = timed(lambda: None, True, graph)
    print(f"{'graph' if graph else 'eager'}: chain alone {t_chain:7.1f} us")
    for name, fn in (("60 small dependent launches", small_chain), ("9x9 dgrad (c4 direct)", c4), ("4 x 50 MB axpby", bigk)):
        a, b = timed(fn, False, graph), timed(fn, True, graph)
        print(f"   {name:30s} alone {a:7.1f} us   beside the chain {b:7.1f} us  (sum {a + t_chain:7.1f})")
print("sync flags", sync2[:2].tolist())
