"""End-to-end parity of the HIP path (tatt_amd.model.tsrn.TSRN_TL_TRANS on cuda:0) against
 (1) the golden vectors generated from the real reference modules (fp32 and fp64 runs), and
 (2) the CPU oracle on the same seeded inputs (incl. dropout through the shared counter-based masks),
plus size-independent properties at the BASELINE batch size.  GPU only."""
import numpy as np
import pytest
import torch

from oracle import procgen, tatt_ref as R
from tests.util import check_against, check_grads, is_buffer, leafify, load_golden, make_state, rel_l2

pytestmark = pytest.mark.gpu

STAGE_TOL = 6e-5     # see tests/test_oracle_golden.py: STN tiny-batch BN + grid_sample amplification in train mode


def build(K=37, seed=0):
    from tatt_amd.model.tsrn import TSRN_TL_TRANS
    m = TSRN_TL_TRANS(text_embedding=K)
    m.load_state_dict(procgen.fill_state(m.state_dict(), seed=seed), strict=True)
    return m.cuda()


def nchw(t, B, H=16, W=64):
    return t.view(B, H, W, -1).permute(0, 3, 1, 2)


def stages(ctx, B):
    tp = ctx["tp"]
    d = {"block1": nchw(ctx["b1"], B), "tp_map": nchw(ctx["tp_map"], B),
         "query_pos": tp[8].view(B, 1024, 64).permute(1, 0, 2), "memory": tp[16].view(B, 26, 64).permute(1, 0, 2),
         "pre_sigmoid": nchw(ctx["tail"][6], B, 32, 128), "block7_in": nchw(ctx["tail"][0], B)}
    for i, s in enumerate(ctx["srbs"]):
        if i > 0:
            d[f"block{i + 1}"] = nchw(s[0], B)
    d["block6"] = d.pop("block7_in")
    return d


@pytest.mark.parametrize("B", [1, 2, 5])
def test_eval_forward_vs_reference_golden(B):
    g = load_golden(f"tsrn_K37_B{B}_eval.npz")
    m = build().eval()
    _, lr, tp = procgen.make_inputs(B, 37, seed=0)
    with torch.no_grad():
        sr, ctx = m.engine().forward(lr.cuda(), tp.cuda(), False, 0)
        sr2 = m(lr.cuda(), tp.cuda())
    assert torch.equal(sr, sr2)
    check_against(g, "sr", sr, 2e-5, 2e-5)
    st = stages(ctx, B)
    for k in ["block1", "query_pos", "memory", "tp_map", "block2", "block6", "pre_sigmoid"]:
        check_against(g, k, st[k], 2e-5)


def stn_scale(n):
    """B=64 gradient pins: the STN head (ReLU / max-pool decisions, x64 pixel scaling) gets 4x the tolerance of everything else."""
    return 4.0 if n.startswith("stn_head.") else 1.0


@pytest.mark.parametrize("rot,prec", [(False, "fp32"), (True, "fp32"), (False, "f32x3")])
def test_config2_and_config4_at_B64_vs_reference_golden(rot, prec):
    """BASELINE config 2 at its own size -- B = 64, train mode -- against goldens made by the real reference (fp32 and fp64 runs):
    output, every stage incl. the 64-step recurrent positional encoding (its chain length IS the batch size), control
    points, per-sample loss, all 231 gradient norms, twelve full gradients, BatchNorm statistics.  At this batch the train-mode
    BatchNorms are well conditioned, so this is the tight train-mode gradient pin: 1.5e-3 of the reference's fp64 run (STN head,
    whose ReLU / max-pool decisions and x64 pixel scaling amplify fp32 rounding, 6e-3).  rot = BASELINE config 4: the same
    with LR / HR pre-deformed by the reference's own rotation (model/__init__.py:4-30, +-10 degrees) through tatt_rotate_fwd,
    asserting the rotated input, control points and the TPS-rectified image.  prec = "f32x3": the same pins, unchanged, for the
    opt-in mode that forms the fp32 products of the trunk's 3x3 convolutions from three bf16 pieces per operand (TATT_F32X3)."""
    B, K = 64, 37
    g = load_golden(f"tsrn_K37_B64_{'rot_' if rot else ''}train.npz")
    m = build(K).train()
    m.dropout_p = 0.0
    m.matmul_precision = prec
    hr, lr, tp = [t.cuda() for t in procgen.make_inputs(B, K, seed=0)]
    eng = m.engine()
    if rot:
        arc = torch.from_numpy(g["arc"]).cuda()
        d_lr, d_hr = torch.empty_like(lr), torch.empty_like(hr)
        eng.L.rotate_fwd(lr, arc, d_lr, B, 4, 16, 64)
        eng.L.rotate_fwd(hr, arc, d_hr, B, 4, 32, 128)
        lr, hr = d_lr, d_hr
        check_against(g, "rot_lr", lr, 2e-6)
    sr, ctx = eng.forward(lr, tp, True, 0, p_drop=0.0)
    st = stages(ctx, B)
    check_against(g, "ctrl_points", ctx["stn"][5].view(B, 20, 2), 2e-5)
    check_against(g, "f64/ctrl_points", ctx["stn"][5].view(B, 20, 2), 2e-5)
    check_against(g, "tps_out", nchw(ctx["xs"], B), STAGE_TOL)
    check_against(g, "f64/tps_out", nchw(ctx["xs"], B), STAGE_TOL)
    for k in ["block1", "query_pos", "memory", "tp_map", "block2", "block4", "block6", "pre_sigmoid"]:
        check_against(g, k, st[k], STAGE_TOL if k != "query_pos" else 2e-6)
    check_against(g, "f64/query_pos", st["query_pos"], 2e-6)
    check_against(g, "f64/sr", sr, 2e-5, 5e-5)
    check_against(g, "sr", sr, 2e-5, 5e-5)
    L = eng.L
    per = torch.empty(B, device="cuda")
    L.image_loss_fwd(sr, hr, per, B, 4, 32, 128, 1.0, 1e-4)
    np.testing.assert_allclose(per.cpu().numpy(), g["image_loss"], rtol=2e-5)
    dsr = torch.empty_like(sr)
    L.image_loss_bwd(sr, hr, None, 100.0 / B, dsr, B, 4, 32, 128, 1.0, 1e-4)
    G = {n: torch.zeros_like(q) for n, q in m.named_parameters()}
    eng.backward(ctx, dsr, G)
    from tatt_amd.engine import unused_param
    check_grads(g, lambda n: None if unused_param(n) else G[n].cpu(), rtol64=1.5e-3, rtol32=1.5e-3, scale=stn_scale)
    sd = m.state_dict()
    for k in [k for k in g if k.startswith("bn:")]:
        check_against(g, k, sd[k[3:]], 1e-5)


def test_config4_rotated_small_batch_vs_reference_golden():
    """BASELINE config 4 at B = 4 (full tensors in the fixture): rotated LR -> STN head -> control points -> TPS grid -> rectified
    image -> SR output, train mode, against the real reference (model/stn_head.py:92-106, model/tps_spatial_transformer.py:97-112)."""
    B, K = 4, 37
    g = load_golden("tsrn_K37_B4_rot_train.npz")
    m = build(K).train()
    hr, lr, tp = [t.cuda() for t in procgen.make_inputs(B, K, seed=0)]
    eng = m.engine()
    arc = torch.from_numpy(g["arc"]).cuda()
    d_lr = torch.empty_like(lr)
    eng.L.rotate_fwd(lr, arc, d_lr, B, 4, 16, 64)
    np.testing.assert_allclose(d_lr.cpu().numpy(), g["rot_lr"], atol=2e-5)   # bilinear weights from fp32 coordinates (same bound as the oracle pin)
    sr, ctx = eng.forward(d_lr, tp, True, 0, p_drop=0.0)
    ctrl = ctx["stn"][5].view(B, 20, 2)
    np.testing.assert_allclose(ctrl.cpu().numpy(), g["f64/ctrl_points"], atol=2e-5)
    # the rotation must actually move the control points away from the identity grid the STN bias encodes
    ident = m.state_dict()["stn_head.stn_fc2.bias"].view(20, 2).cpu()
    assert float((ctrl.cpu() - ident).abs().max()) > 1e-3
    check_against(g, "f64/tps_out", nchw(ctx["xs"], B), 1e-4)
    check_against(g, "tps_out", nchw(ctx["xs"], B), 1e-4)
    check_against(g, "f64/sr", sr, 1e-4, 2e-4)


@pytest.mark.parametrize("K,B", [(37, 2), (37, 5), (2448, 2)])
def test_train_step_vs_reference_golden(K, B):
    g = load_golden(f"tsrn_K{K}_B{B}_train.npz")
    m = build(K).train()
    m.dropout_p = 0.0                                     # parity is defined with dropout off (goldens too)
    hr, lr, tp = procgen.make_inputs(B, K, seed=0)
    sr = m(lr.cuda(), tp.cuda())
    # tight against the reference's fp64 run; loose against its fp32 run, whose own rounding noise (tiny-batch
    # BatchNorm in the STN head, then x64 pixel scaling in grid_sample) is several times larger than ours
    # B=2: BatchNorm1d over two samples in the STN head is maximally ill-conditioned (xhat = +-1)
    check_against(g, "f64/sr", sr, 1e-4 if B == 2 else 2e-5, 3e-4 if B == 2 else 5e-5)
    check_against(g, "sr", sr, 2e-4, 5e-4)
    from tatt_amd.loss.image_loss import ImageLoss
    per = ImageLoss(gradient=True, loss_weight=[1, 1e-4])(sr, hr.cuda())
    np.testing.assert_allclose(per.detach().cpu().numpy(), g["image_loss"], rtol=5e-5)
    loss = per.mean() * 100
    loss.backward()
    grads = {n: p.grad for n, p in m.named_parameters()}
    # Train-mode gradients are ill-conditioned at these tiny batches (tools: /tmp-style sensitivity run recorded in
    # DESIGN.md "Tolerances": a 1e-7 relative input perturbation moves them by 3.5e-4 at B=2, 3e-5 at B=5, and a single
    # ReLU/max-pool decision flip in the STN head moves all STN gradients by ~5e-3), so fp32 implementations agree
    # only to ~1e-2 (B=2) / ~6e-3 (B=5).  The tight gradient pin is test_eval_mode_backward_vs_oracle_fp64 below.
    r64 = 1e-2 if B == 2 else 6e-3
    check_grads(g, lambda n: None if grads[n] is None else grads[n].cpu(), rtol64=r64, rtol32=2 * r64)
    sd = m.state_dict()
    for k in [k for k in g if k.startswith("bn:")]:
        check_against(g, k, sd[k[3:]], 2e-5)


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_train_fwd_bwd_vs_oracle_with_stages(p):
    """B=4 against the oracle, stage by stage, with and without dropout (shared hash masks)."""
    B, K, seed = 4, 37, 5
    state = make_state(K, seed=1)
    m = build(K, seed=1).train()
    hr, lr, tp = procgen.make_inputs(B, K, seed=2)
    sr, ctx = m.engine().forward(lr.cuda(), tp.cuda(), True, seed, p_drop=p)
    # the oracle runs in fp64 here: it is then the rounding-free statement of the reference algorithm (an fp32 CPU
    # run of the same code is itself ~8e-3 off on the STN gradients, measured with tests/debug_grads.py)
    P = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in leafify(state).items()}
    P = {k: (v.detach().requires_grad_(True) if v.is_floating_point() and not is_buffer(k) else v) for k, v in P.items()}
    lr, hr, tp = lr.double(), hr.double(), tp.double()
    inter = {}
    sr_o = R.tsrn_forward(P, lr, tp, training=True, drop=R.HashDropout(p, seed) if p > 0 else None, inter=inter)
    st = stages(ctx, B)
    ctrl = ctx["stn"][5].view(B, 20, 2)
    assert rel_l2(ctrl.cpu().numpy(), inter["ctrl_points"].detach().numpy()) < 2e-5
    for k in ["block1", "query_pos", "memory", "tp_map", "block2", "block4", "block6", "pre_sigmoid"]:
        r = rel_l2(st[k].cpu().numpy(), inter[k].detach().numpy())
        assert r < 2e-5, f"{k}: {r:.3e}"
    assert float((sr.cpu() - sr_o.detach()).abs().max()) < 2e-5
    # backward through the HIP loss kernel and the engine
    L = m.engine().L
    dsr = torch.empty_like(sr)
    L.image_loss_bwd(sr, hr.float().cuda(), None, 100.0 / B, dsr, B, 4, 32, 128, 1.0, 1e-4)
    G = {n: torch.zeros_like(q) for n, q in m.named_parameters()}
    m.engine().backward(ctx, dsr, G)
    (R.image_loss(sr_o, hr).mean() * 100).backward()
    # The same oracle in fp32 measures how far fp32 arithmetic itself lands from the fp64 statement on THIS input: a
    # ReLU / max-pool / bilinear-cell decision of the STN head that sits on a knife edge flips for every fp32
    # implementation alike (seen at B=4: GPU and fp32 CPU both 9.4e-3 from fp64 on stn_convnet.2, 1e-7 from each other
    # stage by stage, tests/debug_grads.py).  The kernels are held to 3e-3, or to the fp32 oracle's own distance.
    P32 = {k: (v.detach().float().requires_grad_(True) if v.is_floating_point() and not is_buffer(k) else
               (v.float() if v.is_floating_point() else v.clone())) for k, v in leafify(state).items()}
    sr32 = R.tsrn_forward(P32, lr.float(), tp.float(), training=True, drop=R.HashDropout(p, seed) if p > 0 else None)
    (R.image_loss(sr32, hr.float()).mean() * 100).backward()
    tot = torch.sqrt(sum((P[n].grad.double() ** 2).sum() for n in G if P[n].grad is not None)).item()
    bad = []
    for n in G:
        ref = P[n].grad
        got = G[n].cpu()
        if ref is None:
            assert float(got.abs().max()) == 0.0, n
            continue
        err = (got.double() - ref.double()).norm().item()
        err32 = (P32[n].grad.double() - ref.double()).norm().item()
        if err > max(3e-3 * ref.double().norm().item(), 1.5 * err32) + 1e-6 * tot:   # train mode at B=4: condition number ~1e3
            bad.append((n, err, err32, ref.norm().item()))
    assert not bad, f"gradient mismatches vs oracle (name, err, fp32-oracle err, |ref|): {bad[:8]}"


@pytest.mark.parametrize("prec", ["fp32", "f32x3"])
def test_eval_mode_backward_vs_oracle_fp64(prec):
    """The tight gradient pin: in eval mode (BatchNorm on running statistics, no STN) the network is well conditioned,
    so every backward kernel of the trunk / TP interpreter / RPE recurrence must match the fp64 oracle to ~1e-4 -- in the
    default fp32-MFMA mode and, to the same bound, in the opt-in three-piece bf16 split mode (TATT_F32X3)."""
    B, K = 3, 37
    state = make_state(K, seed=3)
    m = build(K, seed=3).eval()
    m.matmul_precision = prec
    hr, lr, tp = procgen.make_inputs(B, K, seed=7)
    sr, ctx = m.engine().forward(lr.cuda(), tp.cuda(), False, 0)
    L = m.engine().L
    dsr = torch.empty_like(sr)
    L.image_loss_bwd(sr, hr.cuda(), None, 100.0 / B, dsr, B, 4, 32, 128, 1.0, 1e-4)
    G = {n: torch.zeros_like(q) for n, q in m.named_parameters()}
    m.engine().backward(ctx, dsr, G)
    P = {k: (v.double().requires_grad_(True) if v.is_floating_point() and not is_buffer(k) else
             (v.double() if v.is_floating_point() else v.clone())) for k, v in state.items()}
    sr_o = R.tsrn_forward(P, lr.double(), tp.double(), training=False)
    assert float((sr.cpu() - sr_o.detach()).abs().max()) < 1e-5
    (R.image_loss(sr_o, hr.double()).mean() * 100).backward()
    tot = torch.sqrt(sum((P[n].grad ** 2).sum() for n in G if P[n].grad is not None)).item()
    bad = []
    for n in G:
        ref, got = P[n].grad, G[n].cpu().double()
        if ref is None:
            assert float(got.abs().max()) == 0.0, n
            continue
        err = (got - ref).norm().item()
        if err > 2e-4 * ref.norm().item() + 1e-7 * tot:
            bad.append((n, err / max(ref.norm().item(), 1e-30), ref.norm().item()))
    assert not bad, f"eval-mode gradient mismatches vs fp64 oracle: {bad[:10]}"


def test_state_dict_contract_and_buffers():
    m = build()
    sd = m.state_dict()
    assert len(sd) == 304
    assert sum(q.numel() for q in m.parameters()) == 7608334
    ref = make_state(37)
    for k in ref:
        assert tuple(sd[k].shape) == tuple(ref[k].shape), k
    # default call used by ptflops in the reference (interfaces/base.py:184-186): text_emb=None, B=1, eval
    with torch.no_grad():
        y = m.eval()(torch.rand(1, 4, 16, 64).cuda())
    assert tuple(y.shape) == (1, 4, 32, 128) and torch.isfinite(y).all()


@pytest.mark.parametrize("B", [64, 97])
def test_full_batch_properties(B):
    """BASELINE config 2 size (B=64) and a ragged larger batch (97): finite outputs in (0,1); per-sample independence of the eval path
    (sample i of a batch == the same sample in a batch of 1 is NOT expected because the RPE recurs over the
    batch axis — SURVEY §8a a7 — so instead check prefix consistency: the forward direction of the RPE makes
    query_pos of sample 0 depend on nothing else)."""
    m = build().train()
    hr, lr, tp = procgen.make_inputs(B, 37, seed=4)
    sr = m(lr.cuda(), tp.cuda())
    assert tuple(sr.shape) == (B, 4, 32, 128)
    assert torch.isfinite(sr).all() and float(sr.min()) > 0 and float(sr.max()) < 1
    from tatt_amd.loss.image_loss import ImageLoss
    loss = ImageLoss(gradient=True, loss_weight=[1, 1e-4])(sr, hr.cuda()).mean() * 100
    loss.backward()
    for n, q in m.named_parameters():
        if q.grad is not None:
            assert torch.isfinite(q.grad).all(), n
    # linearity of the backward pass in dsr: grads(2*dsr) == 2*grads(dsr)
    eng = m.engine()
    sr2, ctx = eng.forward(lr.cuda(), tp.cuda(), True, 0, p_drop=0.0)
    d = torch.randn_like(sr2)
    G1 = {n: torch.zeros_like(q) for n, q in m.named_parameters()}
    G2 = {n: torch.zeros_like(q) for n, q in m.named_parameters()}
    eng.backward(ctx, d, G1)
    eng.backward(ctx, 2 * d, G2)
    for n in ["block1.0.weight", "block4.conv2.weight", "infoGen.init_factor.weight", "stn_head.stn_fc2.weight"]:
        assert rel_l2((2 * G1[n]).cpu().numpy(), G2[n].cpu().numpy()) < 1e-5, n


def test_config5_at_its_own_size_B64_K2448():
    """BASELINE config 5 (K = 2448 text-prior classes) at the per-GPU batch it is quoted on: fc_in is a [1664,2448] x [2448,64] product
    and its weight gradient a 1664-row reduction into [64,2448].  Text-branch stages against the fp64 oracle, the engine's own fc_in
    forward / weight-gradient launches at that size against fp64 products, and finiteness + linearity of the whole backward pass."""
    B, K = 64, 2448
    state = make_state(K, seed=2)
    m = build(K, seed=2).train()
    hr, lr, tp = procgen.make_inputs(B, K, seed=6)
    eng = m.engine()
    sr, ctx = eng.forward(lr.cuda(), tp.cuda(), True, 0, p_drop=0.0)
    assert torch.isfinite(sr).all()
    P = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in state.items()}
    inter = {}
    with torch.no_grad():
        sr_o = R.tsrn_forward(P, lr.double(), tp.double(), training=True, inter=inter, fused_rnn=True)
    st = stages(ctx, B)
    for k in ["memory", "query_pos", "tp_map", "block2"]:
        r = rel_l2(st[k].cpu().numpy(), inter[k].detach().numpy())
        assert r < 2e-5, f"{k}: {r:.3e}"
    assert float((sr.cpu() - sr_o).abs().max()) < 5e-5
    # fc_in at 1664 rows: forward and dW / db through the engine's launches
    tpr, t0 = ctx["tp"][0], ctx["tp"][1]
    W, b_ = state["infoGen.fc_in.weight"].double(), state["infoGen.fc_in.bias"].double()
    ref = tpr.cpu().double() @ W.t() + b_
    assert rel_l2(t0.cpu().numpy(), ref.numpy()) < 2e-5
    g = torch.Generator().manual_seed(3)
    dt0 = torch.randn(B * 26, 64, generator=g)
    dW, db = torch.zeros(64, K, device="cuda"), torch.zeros(64, device="cuda")
    eng._w_keep, eng._w_pending, eng._w_defer = [], set(), []
    dt0c = dt0.cuda()
    eng.linear_dwb(dt0c, tpr, dW, db, B * 26, 64, K)
    eng._w_join()
    torch.cuda.synchronize()
    assert rel_l2(dW.cpu().numpy(), (dt0.double().t() @ tpr.cpu().double()).numpy()) < 5e-5
    assert rel_l2(db.cpu().numpy(), dt0.double().sum(0).numpy()) < 5e-5
    # whole backward pass: finite, linear in dsr
    d = torch.randn_like(sr)
    G1 = {n: torch.zeros_like(q) for n, q in m.named_parameters()}
    G2 = {n: torch.zeros_like(q) for n, q in m.named_parameters()}
    eng.backward(ctx, d, G1)
    eng.backward(ctx, 2 * d, G2)
    for n, q in G1.items():
        assert torch.isfinite(q).all(), n
    for n in ["infoGen.fc_in.weight", "infoGen.fc_in.bias", "infoGen.transformer.decoder.layers.0.linear1.weight", "block4.conv2.weight"]:
        assert rel_l2((2 * G1[n]).cpu().numpy(), G2[n].cpu().numpy()) < 1e-5, n


def test_eval_matches_oracle_B16():
    """BASELINE config 1 shape (B=16 eval) against the oracle."""
    B = 16
    m = build().eval()
    _, lr, tp = procgen.make_inputs(B, 37, seed=6)
    with torch.no_grad():
        sr = m(lr.cuda(), tp.cuda())
        ref = R.tsrn_forward(make_state(37), lr, tp, training=False, fused_rnn=True)
    assert float((sr.cpu() - ref).abs().max()) < 2e-5


def test_second_pass_reuses_rpe_chain_of_the_first():
    """forward(rpe_from=ctx1): the recurrent positional encoding depends on the weights only (model/transformer_v2.py:215-221),
    so a second pass of the same step (the TSC term) may reuse the first pass's chain and leave the BPTT to it.  Outputs are
    bit-identical to an independent second pass, the accumulated gradients agree to summation order, and running the backward
    passes in the wrong order is refused."""
    B = 3
    m = build().train()
    eng = m.engine()
    g = torch.Generator().manual_seed(11)
    x1, x2 = torch.rand(B, 4, 16, 64, generator=g).cuda(), torch.rand(B, 4, 16, 64, generator=g).cuda()
    tp = torch.softmax(torch.randn(B, 37, 1, 26, generator=g), 1).cuda()
    d1, d2 = torch.randn(B, 4, 32, 128, generator=g).cuda(), torch.randn(B, 4, 32, 128, generator=g).cuda()

    def run(share):
        buffers = {n: b.clone() for n, b in m.named_buffers()}
        sr1, c1 = eng.forward(x1, tp, True, 0, p_drop=0.1)
        sr2, c2 = eng.forward(x2, tp, True, 1, p_drop=0.1, wpack=c1["wpack"], rpe_from=c1 if share else None)
        G = {n: torch.zeros_like(q) for n, q in m.named_parameters()}
        eng.backward(c2, d2, G)
        eng.backward(c1, d1, G)
        torch.cuda.synchronize()
        for n, b in m.named_buffers():                 # same running statistics for the second run
            b.copy_(buffers[n])
        return sr1.clone(), sr2.clone(), G
    a1, a2, Ga = run(False)
    b1, b2, Gb = run(True)
    assert torch.equal(a1, b1) and torch.equal(a2, b2)
    for n in Ga:
        ref, got = Ga[n].double().cpu().numpy(), Gb[n].double().cpu().numpy()
        assert rel_l2(got, ref) <= 2e-5, (n, rel_l2(got, ref))
    rpe = [n for n in Ga if "gru_encoding" in n]
    assert rpe and all(float(Gb[n].abs().sum()) > 0 for n in rpe)
    # wrong order: the lender's backward before the borrower's
    sr1, c1 = eng.forward(x1, tp, True, 0)
    sr2, c2 = eng.forward(x2, tp, True, 1, wpack=c1["wpack"], rpe_from=c1)
    with pytest.raises(RuntimeError, match="must run its backward first"):
        eng.backward(c1, d1, {n: torch.zeros_like(q) for n, q in m.named_parameters()})
    torch.cuda.synchronize()


def test_mask_false_three_channel_network_vs_oracle():
    """mask=False (reference --mask off: three image channels, model/tsrn.py:586): the engine runs it on the four-channel kernels
    through zero-padded fourth channels.  Train-mode forward (STN on) and eval-mode forward + every gradient against the fp64
    oracle, which takes the three-channel parameters as they are; parity unpinned by a reference-made golden (the fixtures are
    mask=True), pinned through the oracle that those fixtures pin."""
    from tatt_amd.model.tsrn import TSRN_TL_TRANS
    B, K = 3, 37
    m = TSRN_TL_TRANS(text_embedding=K, mask=False)
    state = procgen.fill_state(m.state_dict(), seed=5)
    m.load_state_dict(state, strict=True)
    m = m.cuda()
    assert m.block1[0].weight.shape == (64, 3, 9, 9) and getattr(m, "block8")[1].weight.shape == (3, 64, 9, 9)
    hr, lr, tp = procgen.make_inputs(B, K, seed=9)
    hr, lr = hr[:, :3].contiguous(), lr[:, :3].contiguous()
    P64 = lambda: {k: (v.double().requires_grad_(True) if v.is_floating_point() and not is_buffer(k) else
                       (v.double() if v.is_floating_point() else v.clone())) for k, v in state.items()}
    # train mode: forward only (dropout off), STN head + TPS on three channels
    m.train()
    m.dropout_p = 0.0
    sr = m(lr.cuda(), tp.cuda())
    assert sr.shape == (B, 3, 32, 128)
    sr_o = R.tsrn_forward(P64(), lr.double(), tp.double(), training=True)
    assert float((sr.detach().cpu() - sr_o.detach()).abs().max()) < 1e-4          # B = 3: tiny-batch BatchNorm in the STN head
    # eval mode: forward + all gradients (the tight pin of test_eval_mode_backward_vs_oracle_fp64)
    m.load_state_dict(state, strict=True)            # the train-mode pass above moved the BatchNorm running statistics
    m.eval()
    for p_ in m.parameters():
        p_.grad = None
    sr = m(lr.cuda(), tp.cuda())
    from tatt_amd.loss.image_loss import ImageLoss
    (ImageLoss(gradient=True, loss_weight=[1, 1e-4])(sr, hr.cuda()).mean() * 100).backward()
    P = P64()
    sr_o = R.tsrn_forward(P, lr.double(), tp.double(), training=False)
    assert float((sr.detach().cpu() - sr_o.detach()).abs().max()) < 1e-5
    (R.image_loss(sr_o, hr.double()).mean() * 100).backward()
    tot = torch.sqrt(sum((P[n].grad ** 2).sum() for n, _ in m.named_parameters() if P[n].grad is not None)).item()
    bad = []
    for n, p_ in m.named_parameters():
        ref = P[n].grad
        if ref is None:
            assert p_.grad is None or float(p_.grad.abs().max()) == 0.0, n
            continue
        assert p_.grad is not None and p_.grad.shape == ref.shape, n
        err = (p_.grad.cpu().double() - ref).norm().item()
        if err > 2e-4 * ref.norm().item() + 1e-7 * tot:
            bad.append((n, err / max(ref.norm().item(), 1e-30)))
    assert not bad, f"mask=False eval-mode gradient mismatches vs fp64 oracle: {bad[:10]}"


@pytest.mark.parametrize("prec", ["fp32", "f32x3", "bf16"])
def test_pass_is_reproducible_bit_for_bit(prec):
    """The same B = 64 forward + backward pass four times: every gradient bit-identical.  The pass forks five streams (RPE chain, text
    branch, two weight-gradient streams); nothing in it sums in a run-dependent order.  Before the library was built without
    packed-fp32 VALU code (csrc/Makefile, NOPK) the f32x3 and bf16 passes failed this in about half of the runs: dec_bwd_kernel and
    tps_sample_bwd_k returned wrong lanes beside the split-product weight-gradient kernels (tools/probe_determinism.py)."""
    B, K = 64, 37
    m = build(K).train()
    m.dropout_p = 0.0
    m.matmul_precision = prec
    hr, lr, tp = [t.cuda() for t in procgen.make_inputs(B, K, seed=0)]
    eng = m.engine()
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    ref = None
    for r in range(4):
        m.load_state_dict(sd0)
        sr, ctx = eng.forward(lr, tp, True, 0, p_drop=0.0)
        dsr = torch.empty_like(sr)
        eng.L.image_loss_bwd(sr, hr, None, 100.0 / B, dsr, B, 4, 32, 128, 1.0, 1e-4)
        G = {n: torch.zeros_like(q) for n, q in m.named_parameters()}
        eng.backward(ctx, dsr, G)
        torch.cuda.synchronize()
        if ref is None:
            ref = (sr.clone(), G)
            continue
        assert torch.equal(sr, ref[0]), f"pass {r}: output differs from pass 0"
        bad = [n for n in G if not torch.equal(G[n], ref[1][n])]
        assert not bad, f"pass {r}: {len(bad)} gradients differ from pass 0, e.g. {bad[:4]}"
