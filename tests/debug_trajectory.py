"""One-off GPU check (not collected by pytest): the loss trajectory of N consecutive optimisation steps of SRTrainer against the
CPU oracle's train_step_config2 on the same data, dropout off.  Training amplifies rounding differences (Adam turns
noise-level gradients into +-lr steps), so the curves drift apart slowly; they must not jump apart.
python tests/debug_trajectory.py [steps] [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import procgen, tatt_ref as R
from tests.util import make_state
from tests.test_gpu_trainer import build, is_buffer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
state = make_state(37, seed=2)
hr, lr, tp = procgen.make_inputs(B, 37, seed=3)
from tatt_amd.trainer import SRTrainer
curves = {}
for graph in (False, True):
    m = build(state)
    m.dropout_p = 0.0
    tr = SRTrainer(m, use_graph=graph)
    curves["gpu graph" if graph else "gpu eager"] = [float(tr.step(lr.cuda(), hr.cuda(), tp.cuda()).mean()) * 100 for _ in range(N)]
for dt, name in ((torch.float32, "cpu fp32"), (torch.float64, "cpu fp64")):
    P = {k: (v.detach().clone().to(dt).requires_grad_(True) if v.is_floating_point() and not is_buffer(k) else (v.detach().clone().to(dt) if v.is_floating_point() else v.clone()))
         for k, v in state.items()}
    st = {}
    curves[name] = [float(R.train_step_config2(P, lr.to(dt), hr.to(dt), tp.to(dt), st, i + 1)[0]) for i in range(N)]
print("step " + " ".join(f"{k:>12s}" for k in curves))
for i in range(N):
    print(f"{i + 1:4d} " + " ".join(f"{curves[k][i]:12.5f}" for k in curves))
ref = curves["cpu fp64"]
for k in curves:
    print(k, "max rel dev from fp64:", max(abs(a - b) / b for a, b in zip(curves[k], ref)))
