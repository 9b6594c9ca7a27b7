"""Pins the CPU oracle (oracle/tatt_ref.py) to golden vectors produced by the REAL reference modules
(tests/golden/make_golden.py, run where /root/reference exists).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import procgen, tatt_ref as R
from tests.util import check_against, check_grads, leafify, load_golden, make_state

TOL = 2e-5      # fp32 CPU vs fp32 CPU, differently ordered reductions
# Gradients: two fp32 CPU implementations of the same math (reference modules vs this oracle) already differ by up
# to ~8e-4 rel-L2 at B=2 (tiny-batch BatchNorm in the STN head, cancelling sums such as the PReLU slope); measured
# in this container, see DESIGN.md "Tolerances".
GRAD_RTOL = 1e-3


@pytest.mark.parametrize("B", [1, 2, 5])
def test_eval_forward(B):
    g = load_golden(f"tsrn_K37_B{B}_eval.npz")
    P = make_state(37)
    _, lr, tp = procgen.make_inputs(B, 37, seed=0)
    inter = {}
    with torch.no_grad():
        sr = R.tsrn_forward(P, lr, tp, training=False, inter=inter)
    check_against(g, "sr", sr, TOL, 2e-5)
    for k in ["block1", "query_pos", "memory", "tp_map", "block2", "block6", "block7", "pre_sigmoid"]:
        check_against(g, k, inter[k], TOL)


@pytest.mark.parametrize("K,B", [(37, 2), (37, 5), (2448, 2)])
def test_train_forward_backward_step(K, B):
    g = load_golden(f"tsrn_K{K}_B{B}_train.npz")
    P = leafify(make_state(K))
    hr, lr, tp = procgen.make_inputs(B, K, seed=0)
    inter = {}
    sr = R.tsrn_forward(P, lr, tp, training=True, inter=inter)
    check_against(g, "sr", sr, TOL, 1e-4)     # train mode: tiny-batch BN statistics amplify rounding
    check_against(g, "ctrl_points", inter["ctrl_points"], TOL)
    # the STN's tiny-batch BatchNorm + the x64 pixel scaling of grid_sample amplify fp32 rounding of the control
    # points (6e-6 rel) into ~1.5e-5 rel on the rectified image; everything downstream inherits it
    for k in ["tps_out", "block1", "query_pos", "memory", "tp_map", "block3", "block7", "pre_sigmoid"]:
        check_against(g, k, inter[k], 6e-5)
    per = R.image_loss(sr, hr)
    np.testing.assert_allclose(per.detach().numpy(), g["image_loss"], rtol=2e-5)
    loss = per.mean() * 100
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=2e-5)
    loss.backward()
    check_grads(g, lambda n: P[n].grad)
    names = [str(n) for n in g["grad_names"]]
    for k in [k for k in g if k.startswith("bn:")]:
        check_against(g, k, P[k[3:]], 1e-5)
    # clip + Adam
    fl = [n for n in names]
    state = {}
    with torch.no_grad():
        total = R.clip_and_adam([P[n] for n in fl], [P[n].grad for n in fl], state, 1)
    np.testing.assert_allclose(float(total), g["grad_total_norm"], rtol=1e-3)
    for k in [k for k in g if k.startswith("adam:")]:
        # first Adam step is ~lr*sign(g): elements whose gradient is rounding noise may flip -> loose tolerance
        check_against(g, k, P[k[5:]], 5e-4)


def test_losses_and_metrics():
    g = load_golden("losses.npz")
    hr, lr, _ = procgen.make_inputs(3, 37, seed=1)
    sr, sr2 = torch.from_numpy(g["noise_sr"]), torch.from_numpy(g["noise_sr2"])
    arc = torch.from_numpy(g["arc"])
    np.testing.assert_allclose(R.image_loss(sr, hr).numpy(), g["image_loss"], rtol=1e-5)
    np.testing.assert_allclose(R.rotate(hr, arc).numpy(), g["rot_hr"], atol=2e-5)
    np.testing.assert_allclose(R.rotate(lr, arc).numpy(), g["rot_lr"], atol=2e-5)
    np.testing.assert_allclose(R.tri_ssim(sr, sr2, hr).item(), g["tri_ssim"], rtol=1e-5)
    np.testing.assert_allclose(R.ssim(sr, hr).item(), g["ssim"], rtol=1e-5)
    np.testing.assert_allclose(R.psnr(sr, hr).item(), g["psnr"], rtol=1e-6)


def test_fused_rnn_matches_loop():
    P = make_state(37)
    _, lr, tp = procgen.make_inputs(2, 37, seed=3)
    with torch.no_grad():
        a = R.tsrn_forward(P, lr, tp, training=False)
        b = R.tsrn_forward(P, lr, tp, training=False, fused_rnn=True)
    assert float((a - b).abs().max()) < 2e-6


def test_tps_buffers_match_module():
    from tatt_amd.model.tsrn import TSRN_TL_TRANS
    sd = TSRN_TL_TRANS().state_dict()
    for k, v in R.tps_buffers().items():
        assert torch.allclose(sd["tps." + k], v, atol=1e-6), k
