import sys, torch
sys.path.insert(0, "/root/repo")
from oracle import procgen
from tests.util import rel_l2
from tatt_amd.model.tsrn import TSRN_TL_TRANS
from tatt_amd.engine import unused_param
for B, seed in ((16, 1), (16, 2), (24, 1), (32, 1)):
    res = []
    for fused in (True, False):
        m = TSRN_TL_TRANS(text_embedding=37)
        m.load_state_dict(procgen.fill_state(m.state_dict(), seed=seed))
        m = m.cuda().train()
        m.dropout_p = 0.0
        eng = m.engine()
        eng.dec_fused = fused
        hr, lr, tp = procgen.make_inputs(B, 37, seed=3)
        sr, ctx = eng.forward(lr.cuda(), tp.cuda(), True, 5)
        G = {n: torch.zeros_like(p) for n, p in m.named_parameters()}
        gen = torch.Generator().manual_seed(0)
        eng.backward(ctx, torch.randn(sr.shape, generator=gen).cuda(), G)
        torch.cuda.synchronize()
        res.append((sr, ctx["tp_map"], G))
    (sr_a, tp_a, Ga), (sr_b, tp_b, Gb) = res
    rs = {n: rel_l2(Ga[n].cpu().numpy(), Gb[n].cpu().numpy()) for n in Ga if not unused_param(n) and n != "stn_head.stn_fc1.0.bias"}
    top = sorted(rs.items(), key=lambda kv: -kv[1])[:6]
    print(B, seed, "tp", rel_l2(tp_a.cpu().numpy(), tp_b.cpu().numpy()), "sr", rel_l2(sr_a.cpu().numpy(), sr_b.cpu().numpy()), [(n, f"{r:.1e}") for n, r in top])
