"""Single-launch RPE forward chain (tatt_rpe_chain_fwd) against the 64 per-step launches: bitwise equality and time.
python tools/micro_rpe_chain.py [B]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tatt_amd.lib import lib
L = lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = lambda *s: torch.randn(*s, device="cuda")
gi = g(2, 64, 1536) * 0.5
whf, whb, bhf, bhb = g(1536, 512) * 0.04, g(1536, 512) * 0.04, g(1536) * 0.1, g(1536) * 0.1
dsh, dsg = (B + 1) * 64 * 512, B * 64 * 1536


def bufs():
    return torch.zeros(2, B + 1, 64, 512, device="cuda"), torch.empty(2, B, 64, 1536, device="cuda"), torch.empty(B, 16, 64, 64, device="cuda")


def steps(hst, gha, qp):
    for s in range(B):
        L.rpe_step_fwd(gi, whf, whb, bhf, bhb, hst[0, s], hst[0, s + 1], gha[0, s], qp, s, B, dsg, dsh)


sync = torch.zeros(L.rpe_chain_sync_words(B), device="cuda", dtype=torch.int32)


def chain(hst, gha, qp):
    L.rpe_chain_fwd(gi, whf, whb, bhf, bhb, hst, gha, qp, B, dsg, dsh, sync)


ref, got = bufs(), bufs()
steps(*ref)
chain(*got)
torch.cuda.synchronize()
print("sync header", sync[:2].tolist(), "tickets per XCD", sync[32:32 * 9:32].tolist())
for name, a, b in zip(("hstep", "gh_all", "qp"), ref, got):
    print(name, "bitwise equal" if torch.equal(a, b) else f"MAX DIFF {float((a - b).abs().max())}")
for name, fn in (("64 step launches", steps), ("one chain launch", chain)):
    for _ in range(2):
        fn(*got)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn(*got)
    e1.record()
    e1.synchronize()
    print(f"{name:18s} {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us per chain = {e0.elapsed_time(e1) / 5 / B * 1e3:6.2f} us per step")
print("sync header after timing", sync[:2].tolist())


# ---- backward chain ----
whT = torch.empty(2, 512, 1536, device="cuda")
L.rpe_whh_transpose(whf, whb, whT)
hst, gha, qp_ = ref
dqp = g(B, 16, 64, 64)
sync2 = torch.zeros(L.rpe_chain_sync_words(B), device="cuda", dtype=torch.int32)


def bbufs():
    return torch.zeros(2, 64, 512, device="cuda"), torch.empty(2, B, 64, 1536, device="cuda"), torch.zeros(2, 64, 1536, device="cuda")


def bsteps(car, dga, dgs):
    for s in reversed(range(B)):
        L.rpe_step_bwd(gi, gha[0, s], bhf, bhb, hst[0, s], dqp, car, dga[0, s + 1] if s + 1 < B else None, dga[0, s], dgs, whT, s, B, dsg, dsh)


def bchain(car, dga, dgs):
    L.rpe_chain_bwd(gi, gha, bhf, bhb, hst, dqp, car, dga, dgs, whT, B, dsg, dsh, sync2)


bref, bgot = bbufs(), bbufs()
bsteps(*bref)
bchain(*bgot)
torch.cuda.synchronize()
print("bwd sync header", sync2[:2].tolist())
for name, a, b in zip(("carry", "dgh_all", "dgi_sum"), bref, bgot):
    print(name, "bitwise equal" if torch.equal(a, b) else f"max diff {float((a - b).abs().max()):.3e} (max |ref| {float(a.abs().max()):.3e})")
for name, fn in (("64 bwd step launches", bsteps), ("one bwd chain launch", bchain)):
    for _ in range(2):
        fn(*bgot)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn(*bgot)
    e1.record()
    e1.synchronize()
    print(f"{name:22s} {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us per chain = {e0.elapsed_time(e1) / 5 / B * 1e3:6.2f} us per step")
print("bwd sync header after timing", sync2[:2].tolist())
