"""Debug helper (not collected): fused vs per-op decoder, all gradient errors.  python tests/debug_dec_fused.py [B] [p]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import procgen
from tests.util import rel_l2
from tatt_amd.model.tsrn import TSRN_TL_TRANS
from tatt_amd.engine import unused_param
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
p_drop = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
K = 37
res = []
for fused in (True, False):
    m = TSRN_TL_TRANS(text_embedding=K)
    m.load_state_dict(procgen.fill_state(m.state_dict(), seed=1))
    m = m.cuda().train()
    m.dropout_p = p_drop
    eng = m.engine()
    eng.dec_fused = fused
    hr, lr, tp = procgen.make_inputs(B, K, seed=3)
    sr, ctx = eng.forward(lr.cuda(), tp.cuda(), True, 5)
    G = {n: torch.zeros_like(p) for n, p in m.named_parameters()}
    gen = torch.Generator().manual_seed(0)
    eng.backward(ctx, torch.randn(sr.shape, generator=gen).cuda(), G)
    torch.cuda.synchronize()
    res.append((sr, ctx["tp_map"], G))
(sr_a, tp_a, Ga), (sr_b, tp_b, Gb) = res
print("tp", rel_l2(tp_a.cpu().numpy(), tp_b.cpu().numpy()), "sr", rel_l2(sr_a.cpu().numpy(), sr_b.cpu().numpy()))
rows = []
for n in Ga:
    if unused_param(n):
        continue
    rows.append((rel_l2(Ga[n].cpu().numpy(), Gb[n].cpu().numpy()), n, float(Gb[n].norm())))
for r, n, nm in sorted(rows, reverse=True)[:25]:
    print(f"{r:.3e} {nm:.3e} {n}")
