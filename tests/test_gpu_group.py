"""tatt_tn_small_group: several small weight-gradient problems in one launch pair, against fp64 and against tatt_tn_small."""
import ctypes

import pytest
import torch

from tests.util import rel_l2

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("R", [70000, 1664, 37])
def test_group_matches_fp64_and_single(R):
    from tatt_amd.lib import lib
    L = lib()
    big, hp, g_in, extra = rnd(R, 192, seed=1).cuda(), rnd(R, 64, seed=2).cuda(), rnd(R, 64, seed=3).cuda(), rnd(R, 40, seed=4).cuda()
    R2 = max(1, R // 3)                                      # problems of a group may have different row counts
    probs = [  # (dy view, lddy, x view, ldx, M, N, R, with bias)
        (big.view(-1)[96:], 192, hp.view(-1)[32:], 64, 96, 32, R, True),
        (big, 192, g_in, 64, 96, 64, R, True),
        (g_in, 64, hp, 64, 64, 64, R, False),
        (extra, 40, g_in, 64, 40, 64, R2, True),
        (hp, 64, extra, 40, 7, 33, R, True),
    ]
    ws = torch.empty(L.tn_small_workspace(96, 64), device="cuda")
    dWs = [torch.ones(M, N + 3, device="cuda") for (_, _, _, _, M, N, _, _) in probs]      # row stride lddw > N
    dbs = [torch.ones(M, device="cuda") if wb else None for (*_, M, _, _, wb) in probs]
    n = len(probs)
    VP, LG, IN = ctypes.c_void_p * n, ctypes.c_long * n, ctypes.c_int * n
    L.tn_small_group(n, VP(*[p[0].data_ptr() for p in probs]), LG(*[p[1] for p in probs]), VP(*[p[2].data_ptr() for p in probs]),
                     LG(*[p[3] for p in probs]), VP(*[d.data_ptr() for d in dWs]), LG(*[d.shape[1] for d in dWs]),
                     VP(*[None if b is None else b.data_ptr() for b in dbs]), IN(*[p[4] for p in probs]), IN(*[p[5] for p in probs]),
                     IN(*[p[6] for p in probs]), ws)
    for (dy, lddy, x, ldx, M, N, Rp, wb), dW, db in zip(probs, dWs, dbs):
        dyv = torch.as_strided(dy, (Rp, M), (lddy, 1)).cpu().double()
        xv = torch.as_strided(x, (Rp, N), (ldx, 1)).cpu().double()
        assert rel_l2(dW[:, :N].cpu().double().numpy(), (1 + dyv.t() @ xv).numpy()) < 2e-5
        assert torch.equal(dW[:, N:].cpu(), torch.ones(M, 3))
        if wb:
            assert rel_l2(db.cpu().double().numpy(), (1 + dyv.sum(0)).numpy()) < 2e-5
        one = torch.ones(M, N + 3, device="cuda")
        L.tn_small(dy, lddy, x, ldx, one, N + 3, None, M, N, Rp, ws)
        assert rel_l2(one[:, :N].cpu().double().numpy(), dW[:, :N].cpu().double().numpy()) < 5e-6
