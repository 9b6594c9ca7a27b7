"""Stand-alone timing of the LSTM recurrence at B = 64, T = 26: chain launch vs per-step launches.  python tools/micro_lstm_chain.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tatt_amd.lib import lib  # noqa: E402

L = lib()
B, T = 64, 26
r = lambda *s: torch.randn(*s, device="cuda") * 0.1
gi, whh, bhh = r(T * B, 2048), [r(1024, 256), r(1024, 256)], [r(1024), r(1024)]
rec, sv = torch.empty(T * B, 512, device="cuda"), torch.empty(T * B, 2, 5, 256, device="cuda")
whhT = torch.stack([w.t().contiguous() for w in whh])
do, dg = r(T * B, 512), torch.empty(T * B, 2048, device="cuda")
sf = torch.zeros(L.lstm256_chain_sync_words(), device="cuda", dtype=torch.int32)
sb = torch.zeros(L.lstm256_chain_sync_words(), device="cuda", dtype=torch.int32)
h, c, dcc = torch.zeros(2, 2, B, 256, device="cuda"), torch.zeros(2, 2, B, 256, device="cuda"), torch.zeros(2, B, 256, device="cuda")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def steps_f():
    for s in range(T):
        L.lstm256_step_fwd(gi, whh[0], whh[1], bhh[0], bhh[1], h[s & 1], c[s & 1], h[(s + 1) & 1], c[(s + 1) & 1], rec, sv, s, T, B)


def steps_b():
    for s in reversed(range(T)):
        L.lstm256_step_bwd(sv, whhT, do, dg, dcc, s, T, B)


tf = timeit(lambda: L.lstm256_chain_fwd(gi, whh[0], whh[1], bhh[0], bhh[1], rec, sv, T, B, sf))
tb = timeit(lambda: L.lstm256_chain_bwd(sv, whhT, do, dg, T, B, sb))
print(f"chain fwd {tf:.1f} us ({tf / T:.2f} per step)  bwd {tb:.1f} us ({tb / T:.2f} per step);  step launches (eager) fwd {timeit(steps_f):.1f}  bwd {timeit(steps_b):.1f};  flags {int(sf[1])} {int(sb[1])}")
