"""The whole training iteration (hipGraph replay, dropout on, CRNN strands beside the SR step) run twice from the same state: parameters
after 4 steps must be bit-identical.  python tools/probe_trainer_determinism.py [fp32|f32x3|bf16] [full] [tsc]
(tsc: the config-3 step; its rotation's backward pass is a fixed-order gather since the end of round 3.)"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import procgen  # noqa: E402
from tatt_amd.data import synthetic_batch  # noqa: E402
from tatt_amd.model.tsrn import TSRN_TL_TRANS  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
full = "full" in sys.argv[2:]
tsc = "tsc" in sys.argv[2:]
arc = (torch.rand(64, generator=torch.Generator().manual_seed(5)) * 2 - 1) * 10.0 / 180.0 * 3.141592653589793
B, K = 64, 37
m0 = TSRN_TL_TRANS(text_embedding=K)
m0.load_state_dict(procgen.fill_state(m0.state_dict(), seed=0), strict=True)
hr, lr, tp = synthetic_batch(B, K, seed=123, device="cuda")
if full:
    from tatt_amd.model.crnn import CRNN
    t0, s0 = CRNN(), CRNN()
res = []
for run in range(3):
    torch.manual_seed(0)
    m = copy.deepcopy(m0).cuda().train()
    m.matmul_precision = prec
    kw = dict(lr=1e-3, betas=(0.5, 0.999), max_norm=0.25, loss_weight=(1.0, 1e-4), use_graph=True, tsc=tsc, strict_graph=True)
    if full:
        from tatt_amd.trainer import TATTTrainer
        te, st = copy.deepcopy(t0).cuda(), copy.deepcopy(s0).cuda()
        te.matmul_precision = st.matmul_precision = prec
        tr = TATTTrainer(m, te, st, lr_tpg=1e-4, **kw)
    else:
        from tatt_amd.trainer import SRTrainer
        tr = SRTrainer(m, **kw)
    losses = [tr.step(lr, hr, tp, **({'arc': arc} if tsc else {})).clone() for _ in range(4)]
    torch.cuda.synchronize()
    tr.check_chains() if hasattr(tr, "check_chains") else None
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    if full:
        sd.update({"student." + k: v.detach().clone() for k, v in st.state_dict().items()})
    res.append((torch.stack(losses), sd))
for r in (1, 2):
    bad = [k for k in res[0][1] if not torch.equal(res[0][1][k], res[r][1][k])]
    print(f"{prec}{' full' if full else ''}{' tsc' if tsc else ''}: run {r} vs run 0: losses equal {torch.equal(res[0][0], res[r][0])}, {len(bad)} of {len(res[0][1])} state tensors differ", bad[:5])
