"""Data-parallel SRTrainer over RCCL on the GPU.  The test box has one GPU and RCCL refuses two ranks per device (and two
processes time-slicing one GPU take minutes per step), so the plumbing - parameter broadcast, bucketed AVG all-reduce
issued from the backward pass, the all-reduce after a hipGraph replay, clip+Adam after the last bucket - is exercised with
a ONE-rank `nccl` group, which takes exactly the code path of N ranks.  The N>1 arithmetic (mean over ranks, identical
replicas) is covered on CPU by tests/test_dp_gloo.py."""
import copy
import os
import socket

import pytest
import torch
import torch.distributed as dist

from oracle import procgen
from tests.util import make_state

pytestmark = pytest.mark.gpu
B, K = 3, 37


def _build(state):
    from tatt_amd.model.tsrn import TSRN_TL_TRANS
    m = TSRN_TL_TRANS(text_embedding=K)
    m.load_state_dict(copy.deepcopy(state), strict=True)
    m = m.cuda()
    m.dropout_p = 0.0
    return m


@pytest.fixture(scope="module")
def one_rank_group():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("graph", [False, True])
def test_dp_path_over_rccl_matches_single_process_step(one_rank_group, graph):
    from tatt_amd.trainer import SRTrainer
    state = make_state(K, seed=11)
    hr, lr, tp = [t.cuda() for t in procgen.make_inputs(B, K, seed=20)]
    tr = SRTrainer(_build(state), use_graph=graph)
    assert tr.dp and tr.world == 1
    ref = SRTrainer(_build(state), use_graph=graph, distributed=False)
    assert not ref.dp
    for _ in range(2):
        tr.step(lr, hr, tp)
        ref.step(lr, hr, tp)
    torch.cuda.synchronize()
    assert torch.equal(tr.flat_g, ref.flat_g), "mean all-reduce over one rank must be the identity"
    assert torch.equal(tr.flat_p, ref.flat_p)
