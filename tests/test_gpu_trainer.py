"""SRTrainer on the GPU: one optimisation step against the oracle's step, and hipGraph replay against eager launches."""
import copy

import numpy as np
import pytest
import torch

from oracle import procgen, tatt_ref as R
from tests.util import is_buffer, make_state, rel_l2

pytestmark = pytest.mark.gpu


def build(state):
    from tatt_amd.model.tsrn import TSRN_TL_TRANS
    m = TSRN_TL_TRANS(text_embedding=37)
    m.load_state_dict(state, strict=True)
    return m.cuda()


def test_one_step_matches_oracle_step():
    from tatt_amd.trainer import SRTrainer
    B = 6
    state = make_state(37, seed=2)
    m = build(state)
    m.dropout_p = 0.0
    tr = SRTrainer(m, use_graph=False)
    hr, lr, tp = procgen.make_inputs(B, 37, seed=3)
    per = tr.step(lr.cuda(), hr.cuda(), tp.cuda())
    # oracle step in fp64 (rounding-free statement of the reference step)
    P = {k: (v.double().requires_grad_(True) if v.is_floating_point() and not is_buffer(k) else
             (v.double() if v.is_floating_point() else v.clone())) for k, v in state.items()}
    loss, total, _ = R.train_step_config2(P, lr.double(), hr.double(), tp.double(), {}, 1)
    assert abs(float(per.mean()) * 100 - float(loss)) < 2e-5 * float(loss)
    assert abs(float(tr.state[0]) - float(total)) < 2e-3 * float(total)          # global grad norm (train-mode conditioning)
    sd = m.state_dict()
    # first Adam step is ~ -lr*sign(g): compare the parameter DELTAS; noise-level gradients may flip sign, so look at
    # the fraction of elements that moved the same way
    for k in ["block3.conv1.weight", "block8.0.conv.weight", "infoGen.transformer.decoder.layers.0.linear2.weight",
              "block5.gru2.gru.weight_hh_l0", "stn_head.stn_convnet.4.0.weight", "block1.0.weight"]:
        d_gpu = (sd[k].cpu().double() - state[k].double()).numpy().ravel()
        d_ref = (P[k].detach() - state[k].double()).numpy().ravel()
        agree = np.mean(np.sign(d_gpu) == np.sign(d_ref))
        assert agree > 0.995, f"{k}: only {agree:.4f} of the Adam updates agree in sign"
        assert abs(np.abs(d_gpu).mean() - np.abs(d_ref).mean()) < 0.02 * np.abs(d_ref).mean(), k
    # parameters that never receive a gradient stay untouched
    for k in ["infoGen.fc_feature_in.weight", "infoGen.transformer.decoder.layers.1.self_attn.in_proj_weight"]:
        assert torch.equal(sd[k].cpu(), state[k]), k
    # parameters are views of the flat buffer and state_dict round-trips
    assert all(p.data_ptr() >= tr.flat_p.data_ptr() for p in m.parameters())


def test_graph_replay_equals_eager():
    from tatt_amd.trainer import SRTrainer
    B = 4
    state = make_state(37, seed=4)
    hr, lr, tp = [t.cuda() for t in procgen.make_inputs(B, 37, seed=5)]
    res = []
    for graph in (False, True):
        m = build(copy.deepcopy(state))
        tr = SRTrainer(m, use_graph=graph)                 # dropout 0.1 stays ON: the device-side seed must advance per replay
        losses = [tr.step(lr, hr, tp).clone() for _ in range(3)]
        torch.cuda.synchronize()
        res.append((tr.flat_p.clone(), torch.stack(losses), int(tr.seed_dev.item()), float(tr.state[2]),
                    {k: v.clone() for k, v in m.named_buffers() if "running_" in k or "num_batches" in k}))
    (p0, l0, s0, n0, b0), (p1, l1, s1, n1, b1) = res
    assert s0 == s1 == 3 and n0 == n1 == 3.0
    # the warm-up pass before capture leaves no trace: BatchNorm running statistics and counters equal the eager path's
    for k in b0:
        assert torch.equal(b0[k], b1[k]), f"buffer {k} differs between eager and graph training (warm-up side effect?)"
    assert int(b0["block2.bn1.num_batches_tracked"]) == 3
    assert torch.equal(l0, l1), "graph replay drew different dropout masks / losses than eager execution"
    assert torch.equal(p0, p1), "parameters after 3 steps differ between graph replay and eager launches"
    assert float((l0[0] - l0[1]).abs().max()) > 0, "losses identical across steps: dropout seed or update not advancing"


def test_tsc_step_matches_oracle_step():
    """Config-3 style step in fp32: L_SR + L_TSC (second SR forward on the rotated input, three rotations, TRI_SSIM),
    against the oracle's fp64 statement of interfaces/super_resolution_eng.py:233-279."""
    from tatt_amd.trainer import SRTrainer
    B = 4
    state = make_state(37, seed=6)
    m = build(state)
    m.dropout_p = 0.0
    tr = SRTrainer(m, use_graph=False, tsc=True)
    hr, lr, tp = procgen.make_inputs(B, 37, seed=7)
    arc = torch.tensor([0.12, -0.05, 0.17, -0.15])
    per = tr.step(lr.cuda(), hr.cuda(), tp.cuda(), arc=arc)
    P = {k: (v.double().requires_grad_(True) if v.is_floating_point() and not is_buffer(k) else
             (v.double() if v.is_floating_point() else v.clone())) for k, v in state.items()}
    loss_img, loss_tssim, total, _, _ = R.train_step_config3(P, lr.double(), hr.double(), tp.double(), arc.double(), {}, 1)
    assert abs(float(per.mean()) * 100 - float(loss_img)) < 2e-5 * float(loss_img)
    assert abs((1 - float(tr.tssim)) * 10 - float(loss_tssim)) < 1e-4 * float(loss_tssim)
    assert abs(float(tr.state[0]) - float(total)) < 5e-3 * float(total)          # global grad norm (train-mode conditioning)
    sd = m.state_dict()
    # BatchNorm statistics were updated twice (two SR forwards per step)
    assert int(sd["block3.bn1.num_batches_tracked"]) == 2
    for k in ["block3.bn1.running_mean", "block7.1.running_var", "stn_head.stn_convnet.4.1.running_mean"]:
        assert rel_l2(sd[k].cpu().double().numpy(), P[k].detach().numpy()) < 1e-4, k
    for k in ["block3.conv1.weight", "block8.0.conv.weight", "block5.gru2.gru.weight_hh_l0", "block1.0.weight"]:
        d_gpu = (sd[k].cpu().double() - state[k].double()).numpy().ravel()
        d_ref = (P[k].detach() - state[k].double()).numpy().ravel()
        assert np.mean(np.sign(d_gpu) == np.sign(d_ref)) > 0.99, k


@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_tsc_batched_passes_equal_sequential_passes(p_drop):
    """The two SR passes of a TSC step as one 2B-row batch (Engine.forward(x2=...): per-pass BatchNorm statistics, dropout seeds, STN head
    and text branch) against the same step with one pass after the other (rpe_from): loss terms, every gradient, BatchNorm running
    statistics after their two updates, num_batches_tracked == 2."""
    from tatt_amd.trainer import SRTrainer
    B = 16                                              # (B = 8 sits at 6e-5 on block1.0.weight: summation order on an ill-conditioned small batch)
    state = make_state(37, seed=9)
    hr, lr, tp = procgen.make_inputs(B, 37, seed=10)
    arc = torch.linspace(-0.17, 0.17, B)
    res = []
    for batched in (True, False):
        m = build(copy.deepcopy(state))
        m.dropout_p = p_drop
        tr = SRTrainer(m, use_graph=False, tsc=True)
        tr.tsc_batched = batched
        per = tr.step(lr.cuda(), hr.cuda(), tp.cuda(), arc=arc)
        torch.cuda.synchronize()
        sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        res.append((per.cpu(), float(tr.tssim), tr.flat_g.cpu().clone(), float(tr.state[0]), sd, {n: g.cpu().clone() for n, g in tr.G.items()}))
    (per_a, ts_a, g_a, n_a, sd_a, G_a), (per_b, ts_b, g_b, n_b, sd_b, G_b) = res
    assert rel_l2(per_a.numpy(), per_b.numpy()) < 1e-6 and abs(ts_a - ts_b) < 1e-6
    assert abs(n_a - n_b) < 1e-5 * n_b
    assert int(sd_a["block3.bn1.num_batches_tracked"]) == 2 and int(sd_a["stn_head.stn_convnet.4.1.num_batches_tracked"]) == 2
    for k in sd_a:
        if "running_" in k:
            assert rel_l2(sd_a[k].double().numpy(), sd_b[k].double().numpy()) < 1e-6, k
    from tatt_amd.engine import unused_param
    for n in G_a:
        if unused_param(n) or n == "stn_head.stn_fc1.0.bias":
            continue
        r = rel_l2(G_a[n].numpy(), G_b[n].numpy())
        assert r < 5e-5, (n, r)


def test_step_normalises_input_layout_and_dtype():
    """step() accepts what a data loader may hand over -- a sliced (non-contiguous) LR batch, a double HR batch, a non-contiguous
    prior view -- and computes the same step as with plain contiguous fp32 tensors (TSC on: the rotation and the second SR pass read
    the LR batch too)."""
    from tatt_amd.trainer import SRTrainer
    B = 3
    state = make_state(37, seed=14)
    hr, lr, tp = procgen.make_inputs(B, 37, seed=15)
    arc = torch.tensor([0.1, -0.12, 0.05])
    out = []
    for odd in (False, True):
        m = build(copy.deepcopy(state))
        m.dropout_p = 0.0
        tr = SRTrainer(m, use_graph=False, tsc=True)
        if odd:
            big = torch.zeros(B, 4, 16, 80)
            big[..., 8:72] = lr
            lr_in = big.cuda()[..., 8:72]                                 # a view with a row stride of 80
            hr_in = hr.double().cuda()
            tp_in = tp.permute(0, 3, 2, 1).contiguous().cuda().permute(0, 3, 2, 1)   # [B,K,1,26] view of a [B,26,1,K] buffer
            assert not lr_in.is_contiguous() and not tp_in.is_contiguous()
        else:
            lr_in, hr_in, tp_in = lr.cuda(), hr.cuda(), tp.cuda()
        per = tr.step(lr_in, hr_in, tp_in, arc=arc)
        torch.cuda.synchronize()
        out.append((per.clone(), tr.tssim.clone(), tr.flat_g.clone()))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    # gradients too, bit for bit: tatt_rotate_bwd is a gather with a fixed summation order (it scattered with float atomics until the
    # end of round 3, the one sum in the library whose order changed from run to run)
    assert torch.equal(out[1][2], out[0][2])


def test_config3_step_bf16_with_tsc_vs_fp64_oracle():
    """BASELINE config 3 as stated: the TSC step (two SR passes, three rotations, TRI_SSIM) with bf16 matrix-core operands,
    against the oracle's fp64 statement of the same step.  Tolerances: SURVEY 8c's bf16 proposal (output-level quantities within
    1e-2 relative, gradient cosine >= 0.999)."""
    from tatt_amd.trainer import SRTrainer
    B = 4
    state = make_state(37, seed=6)
    m = build(state)
    m.dropout_p = 0.0
    m.matmul_precision = "bf16"
    tr = SRTrainer(m, use_graph=False, tsc=True)
    hr, lr, tp = procgen.make_inputs(B, 37, seed=7)
    arc = torch.tensor([0.12, -0.05, 0.17, -0.15])
    per = tr.step(lr.cuda(), hr.cuda(), tp.cuda(), arc=arc)
    P = {k: (v.double().requires_grad_(True) if v.is_floating_point() and not is_buffer(k) else
             (v.double() if v.is_floating_point() else v.clone())) for k, v in state.items()}
    loss_img, loss_tssim, total, _, _ = R.train_step_config3(P, lr.double(), hr.double(), tp.double(), arc.double(), {}, 1)
    l_img, l_ts, gn = float(per.mean()) * 100, (1 - float(tr.tssim)) * 10, float(tr.state[0])
    print(f"bf16+TSC: L_SR {l_img:.5f} vs {float(loss_img):.5f}  L_TSC {l_ts:.5f} vs {float(loss_tssim):.5f}  |g| {gn:.4f} vs {float(total):.4f}")
    assert abs(l_img - float(loss_img)) > 1e-7 * float(loss_img)                # it is not the fp32 path
    assert abs(l_img - float(loss_img)) < 1e-2 * float(loss_img)
    assert abs(l_ts - float(loss_tssim)) < 1e-2 * float(loss_tssim)
    assert abs(gn - float(total)) < 5e-2 * float(total)
    # gradient direction of the TWO-PASS step (both SR passes as one 2B batch, bf16 operands and bf16-stored trunk tensors): SURVEY 8c's
    # bf16 bar, cosine >= 0.999 against the fp64 step's gradient over all parameters
    from tatt_amd.engine import unused_param
    names = [n for n in tr.G if P[n].grad is not None and not unused_param(n)]
    g = torch.cat([tr.G[n].detach().cpu().double().reshape(-1) for n in names])
    g64 = torch.cat([P[n].grad.reshape(-1) for n in names])
    cos = float((g @ g64) / (g.norm() * g64.norm()))
    print(f"bf16+TSC two-pass step: gradient cosine {cos:.6f} over {len(names)} tensors")
    assert len(names) > 200 and cos >= 0.999, cos
    sd = m.state_dict()
    assert int(sd["block3.bn1.num_batches_tracked"]) == 2
    for k in ["block3.bn1.running_mean", "block7.1.running_var"]:
        assert rel_l2(sd[k].cpu().double().numpy(), P[k].detach().numpy()) < 1e-2, k
    for k in ["block3.conv1.weight", "block8.0.conv.weight", "block5.gru2.gru.weight_hh_l0"]:
        d_gpu = (sd[k].cpu().double() - state[k].double()).numpy().ravel()
        d_ref = (P[k].detach() - state[k].double()).numpy().ravel()
        assert np.mean(np.sign(d_gpu) == np.sign(d_ref)) > 0.9, k


def test_tsc_graph_step_runs_and_loss_stablizing_mask():
    from tatt_amd.trainer import SRTrainer
    B = 4
    state = make_state(37, seed=8)
    hr, lr, tp = [t.cuda() for t in procgen.make_inputs(B, 37, seed=9)]
    m = build(state)
    tr = SRTrainer(m, use_graph=True, tsc=True, training_stablize=True)
    for _ in range(3):
        per = tr.step(lr, hr, tp)
    assert torch.isfinite(per).all() and 0.0 < float(tr.tssim) <= 1.0
    assert int((per == 0).sum()) >= 1                                  # the stabiliser dropped at least one sample
    for seed in range(5):
        v = torch.rand(10, generator=torch.Generator().manual_seed(seed))
        ref = R.loss_stablizing(v.clone())
        got = v.cuda() * SRTrainer.loss_stablizing_mask(v.cuda())
        assert torch.equal(got.cpu(), ref)


def test_host_mirrors_tri_ssim_and_rotation_autograd():
    """tatt_amd.utils.ssim_psnr.{TRI_SSIM,SSIM,calculate_psnr} and tatt_amd.model.torch_distortion behave like the
    reference modules they mirror (utils/ssim_psnr.py, model/__init__.py:4-30), including autograd."""
    from tatt_amd.model import torch_distortion
    from tatt_amd.utils.ssim_psnr import SSIM, TRI_SSIM, calculate_psnr
    g = torch.Generator().manual_seed(3)
    a, b, c = (torch.rand(2, 4, 32, 128, generator=g) for _ in range(3))
    arc = torch.tensor([0.1, -0.16])
    ad, bd = a.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = (1 - R.tri_ssim(R.rotate(ad, arc), bd, R.rotate(c.double(), arc))) * 10
    ref.backward()
    ag, bg = a.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    out = (1 - TRI_SSIM()(torch_distortion(ag, arc), bg, torch_distortion(c.cuda(), arc))) * 10
    out.backward()
    assert abs(float(out) - float(ref)) < 1e-5
    assert rel_l2(ag.grad.cpu().double().numpy(), ad.grad.numpy()) < 1e-3
    assert rel_l2(bg.grad.cpu().double().numpy(), bd.grad.numpy()) < 1e-3
    assert abs(float(SSIM()(a.cuda(), c.cuda())) - float(R.ssim(a.double(), c.double()))) < 2e-6
    assert abs(float(calculate_psnr(a.cuda(), c.cuda())) - float(R.psnr(a.double(), c.double()))) < 1e-4


def _crnn(seed, train):
    from tatt_amd.model.crnn import CRNN
    m = CRNN()
    st = procgen.fill_state(m.state_dict(), seed=seed)
    m.load_state_dict(st, strict=True)
    return m.cuda().train(train), st


@pytest.mark.parametrize("graph", [False, True])
def test_full_reference_iteration_matches_oracle(graph):
    """TATTTrainer = the reference's whole train() iteration (teacher + student CRNN priors, SR step, SemanticLoss,
    two-group Adam) against the oracle's fp64 statement of interfaces/super_resolution_eng.py:214-279."""
    from oracle import crnn_ref as C
    from tatt_amd.trainer import TATTTrainer
    B = 3
    state = make_state(37, seed=12)
    m = build(state)
    m.dropout_p = 0.0
    teacher, st_t = _crnn(21, False)
    student, st_s = _crnn(22, True)
    tr = TATTTrainer(m, teacher, student, use_graph=graph)
    hr, lr, tp = procgen.make_inputs(B, 37, seed=13)
    per = tr.step(lr.cuda(), hr.cuda(), tp.cuda())
    torch.cuda.synchronize()
    dbl = lambda st, buf: {k: (v.double().requires_grad_(True) if v.is_floating_point() and not buf(k) else
                               (v.double() if v.is_floating_point() else v.clone())) for k, v in st.items()}
    P_sr = dbl(state, is_buffer)
    P_t = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in st_t.items()}
    P_s = dbl(st_s, lambda k: "running_" in k)
    loss_img, _, loss_tp, total, _ = C.full_train_step(P_sr, P_t, P_s, lr.double(), hr.double(), {}, {}, 1)
    assert abs(float(per.mean()) * 100 - float(loss_img)) < 5e-5 * float(loss_img)
    assert abs(float(tr.loss_tp) * 100 - float(loss_tp)) < 1e-4 * float(loss_tp)
    assert abs(float(tr.state[0]) - float(total)) < 5e-3 * float(total)
    sd_s, sd_t = student.state_dict(), teacher.state_dict()
    for k in ["rnn.1.embedding.weight", "rnn.0.rnn.weight_hh_l0", "cnn.conv6.weight", "cnn.conv3.weight", "cnn.batchnorm2.weight"]:
        d_gpu = (sd_s[k].cpu().double() - st_s[k].double()).numpy().ravel()
        d_ref = (P_s[k].detach() - st_s[k].double()).numpy().ravel()
        assert np.mean(np.sign(d_gpu) == np.sign(d_ref)) > 0.99, k                  # first Adam step ~ -lr_tpg * sign(g)
        assert abs(np.abs(d_gpu).mean() - 1e-4) < 2e-5, k                           # student group runs at lr 1e-4
    for k, v in st_t.items():
        assert torch.equal(sd_t[k].cpu(), v), f"teacher changed: {k}"
    # the graph path warms up once before capture and puts every BatchNorm buffer back: one training step = one update
    assert int(sd_s["cnn.batchnorm4.num_batches_tracked"]) == 1
    assert int(m.state_dict()["block3.bn1.num_batches_tracked"]) == 1
    for k in ("cnn.batchnorm2.running_mean", "cnn.batchnorm6.running_var"):
        assert rel_l2(sd_s[k].cpu().double().numpy(), P_s[k].detach().numpy()) < 1e-4, k
