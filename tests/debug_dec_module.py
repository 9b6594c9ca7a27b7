"""Not collected: prints the largest fused-vs-per-op decoder gradient differences of the module (see test_gpu_dec_fused.py)."""
import sys
import torch
from oracle import procgen
from tests.util import rel_l2
from tatt_amd.model.tsrn import TSRN_TL_TRANS
from tatt_amd.engine import unused_param

B, K = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 37
for rep in range(2):
    res = []
    for fused in (True, False):
        m = TSRN_TL_TRANS(text_embedding=K)
        m.load_state_dict(procgen.fill_state(m.state_dict(), seed=1))
        m = m.cuda().train()
        m.dropout_p = 0.0
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
    rs = sorted(((rel_l2(Ga[n].cpu().numpy(), Gb[n].cpu().numpy()), n, float(Ga[n].norm())) for n in Ga if not unused_param(n)), reverse=True)
    print("rep", rep, [(f"{r:.1e}", n, f"{nm:.2e}") for r, n, nm in rs[:6]])
