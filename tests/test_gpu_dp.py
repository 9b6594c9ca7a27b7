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


def test_graph_dp_is_segmented_at_bucket_boundaries_and_equals_eager_dp(one_rank_group):
    """Data parallel + hipGraph: the step is captured as two segments cut where the SRBs' gradients are complete (the all-reduce of
    tail + SRBs is issued between the two replays and runs while the second segment executes) plus one clip+Adam graph; the result is
    bit-identical to the eager data-parallel step, which issues the same all-reduces from inside the backward pass."""
    from tatt_amd.trainer import SRTrainer
    state = make_state(K, seed=12)
    hr, lr, tp = [t.cuda() for t in procgen.make_inputs(B, K, seed=21)]
    eager = SRTrainer(_build(state), use_graph=False)
    graph = SRTrainer(_build(state), use_graph=True)
    graph.time_comm = True
    for _ in range(3):
        eager.step(lr, hr, tp)
        graph.step(lr, hr, tp)
    torch.cuda.synchronize()
    assert [b for _, b in graph.graph] == [["srbs"], ["infogen", "extra"]] and graph.graph_update is not None
    a, b = graph._bucket_views("srbs")[0], graph._bucket_views("infogen")[0]         # the two collectives tile the gradient buffer
    assert a.data_ptr() == graph.flat_g.data_ptr() and b.data_ptr() == a.data_ptr() + 4 * a.numel() and a.numel() + b.numel() == graph.n
    assert torch.equal(eager.flat_g, graph.flat_g) and torch.equal(eager.flat_p, graph.flat_p)
    assert graph.exposed_comm_ms() >= 0.0 and graph.verify_replicas() == 1


def test_full_iteration_graph_dp_equals_eager_dp(one_rank_group):
    """The same for the reference's whole iteration (student / teacher CRNN strands on their own streams join before the first cut)."""
    from tatt_amd.model.crnn import CRNN
    from tatt_amd.trainer import TATTTrainer
    state = make_state(K, seed=13)
    hr, lr, tp = [t.cuda() for t in procgen.make_inputs(B, K, seed=22)]
    res = []
    for g in (False, True):
        teacher, student = CRNN(), CRNN()
        teacher.load_state_dict(procgen.fill_state(teacher.state_dict(), seed=31))
        student.load_state_dict(procgen.fill_state(student.state_dict(), seed=32))
        tr = TATTTrainer(_build(state), teacher.cuda(), student.cuda(), use_graph=g)
        for _ in range(2):
            tr.step(lr, hr, tp)
        torch.cuda.synchronize()
        res.append((tr.flat_p.clone(), tr.s_p.clone()))
        assert tr.verify_replicas() == 1
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
