"""Data-parallel plumbing on CPU: world_size-2 gloo processes exercise the flat gradient layout and the bucketed
mean all-reduce used by SRTrainer (the RCCL path differs only in the backend's AVG op)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tatt_amd.trainer import BUCKETS, REDUCE_AT, allreduce_mean, assert_replicas_identical, broadcast_buffers, plan_flat_layout


def _shapes():
    from tatt_amd.model.tsrn import TSRN_TL_TRANS
    m = TSRN_TL_TRANS()
    return [(n, p.numel()) for n, p in m.named_parameters()], m.srb_nums


def test_flat_layout_is_in_gradient_production_order():
    shapes, srb = _shapes()
    offs, total, ranges = plan_flat_layout(shapes, srb)
    assert len(offs) == 245 and total >= 7608334 and total % 64 == 0
    # buckets are contiguous, ordered tail -> srbs -> infogen -> head and tile the buffer exactly
    assert ranges[0][0] == 0 and ranges[-1][1] == total
    for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
        assert a1 == b0 and a0 < a1
    name_at = {v: k for k, v in offs.items()}
    assert name_at[0].startswith("block7.")
    assert all(ranges[1][0] <= offs[n] < ranges[1][1] for n, _ in shapes if n.startswith("block4."))
    assert offs["block6.conv1.weight"] < offs["block2.conv1.weight"]          # last SRB's gradients are produced first
    assert ranges[2][0] <= offs["infoGen.init_factor.weight"] < ranges[2][1]
    assert offs["block1.0.weight"] >= ranges[3][0] and offs["stn_head.stn_fc2.bias"] >= ranges[3][0]
    # slots do not overlap
    spans = sorted((o, o + dict(shapes)[n]) for n, o in offs.items())
    assert all(e0 <= s1 for (_, e0), (s1, _) in zip(spans, spans[1:]))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes, srb = _shapes()
    offs, total, ranges = plan_flat_layout(shapes, srb)
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(total, generator=g)
    mine = flat.clone()
    for first, last in REDUCE_AT.values():                # same calls, same order as SRTrainer.on_stage: two collectives per step
        allreduce_mean(flat[ranges[BUCKETS.index(first)][0]:ranges[BUCKETS.index(last)][1]], async_op=True)
    other = torch.randn(total, generator=torch.Generator().manual_seed(100 + (1 - rank)))
    ok = torch.allclose(flat, (mine + other) / 2, atol=1e-6)
    # the replica self-check bench.py runs after its timed steps: passes on the (now identical) reduced buffer, raises on
    # EVERY rank when one replica differs in a single low bit
    ok = ok and assert_replicas_identical(flat) == world
    if rank == 1:
        flat.view(torch.int32)[12345] ^= 1
    try:
        assert_replicas_identical(flat)
        ok = False
    except RuntimeError as e:
        ok = ok and "diverged" in str(e)
    # module buffers: packed per dtype into one broadcast each, rank 0's values land in every rank's own tensors (shapes kept)
    bufs = [torch.full((3, 5), float(rank + 1)), torch.tensor(7 + rank), torch.full((4,), rank + 2, dtype=torch.int64), torch.full((2, 2), 0.5 + rank)]
    ptrs = [b.data_ptr() for b in bufs]
    broadcast_buffers(bufs)
    ok = ok and all(b.data_ptr() == p for b, p in zip(bufs, ptrs))
    ok = ok and torch.equal(bufs[0], torch.full((3, 5), 1.0)) and int(bufs[1]) == 7 and torch.equal(bufs[2], torch.full((4,), 2, dtype=torch.int64))
    ok = ok and torch.equal(bufs[3], torch.full((2, 2), 0.5))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_mean_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
