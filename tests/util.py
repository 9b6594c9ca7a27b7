"""Shared helpers of the parity tests."""
import os

import numpy as np
import torch

from oracle import procgen

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAXS = 4096


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def summarize(t: torch.Tensor):
    f = t.detach().reshape(-1).double().cpu()
    stride = max(1, f.numel() // MAXS)
    return f[::stride][:MAXS].float().numpy(), np.array([f.sum().item(), (f * f).sum().item(), f.numel()], np.float64)


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def check_against(gold, name, t, tol_rel=1e-5, tol_abs=None):
    """Compare tensor t with the golden entry `name` (full tensor, or strided sample + stats)."""
    if name in gold:
        ref = gold[name]
        got = t.detach().float().cpu().numpy().reshape(ref.shape)
        r = rel_l2(got, ref)
        assert r <= tol_rel, f"{name}: rel-L2 {r:.3e} > {tol_rel}"
        if tol_abs is not None:
            m = float(np.abs(got - ref).max())
            assert m <= tol_abs, f"{name}: max-abs {m:.3e} > {tol_abs}"
        return r
    s, st = summarize(t)
    ref_s, ref_st = gold[name + "@sample"], gold[name + "@stats"]
    assert st[2] == ref_st[2], f"{name}: numel {st[2]} != {ref_st[2]}"
    r = rel_l2(s, ref_s)
    assert r <= tol_rel, f"{name}: sample rel-L2 {r:.3e} > {tol_rel}"
    n2 = abs(np.sqrt(st[1]) - np.sqrt(ref_st[1])) / max(np.sqrt(ref_st[1]), 1e-30)
    assert n2 <= tol_rel, f"{name}: norm mismatch {n2:.3e}"
    return r


def make_state(K=37, seed=0):
    """Procedural state dict on CPU with the SR network's keys/shapes (taken from the product module shell,
    which only defines parameter names and shapes)."""
    from tatt_amd.model.tsrn import TSRN_TL_TRANS
    m = TSRN_TL_TRANS(text_embedding=K)
    return procgen.fill_state(m.state_dict(), seed=seed)


def leafify(state):
    return {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not is_buffer(k) else v.clone())
            for k, v in state.items()}


def is_buffer(k):
    return (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked")
            or k.startswith("tps.") or k.endswith("pe.pe"))


# Gradient tolerances (measured in the build container, see DESIGN.md "Tolerances"): the reference's own fp32 run
# differs from its fp64 run by up to 5e-3 rel-L2 on a few tensors (init_factor, PReLU slopes, STN head at B=2),
# so gradients are checked tightly against the fp64 reference run and loosely against the fp32 one.
GRAD_RTOL64 = 1.5e-3
GRAD_RTOL32 = 1e-2


def check_grads(gold, grad_of, rtol64=GRAD_RTOL64, rtol32=GRAD_RTOL32, scale=None):
    """gold: a *_train.npz fixture; grad_of(name) -> tensor or None; scale(name) -> per-tensor multiplier of the tolerances."""
    names = [str(n) for n in gold["grad_names"]]
    total = float(gold["grad_total_norm"])
    for pre, rtol in (("f64/", rtol64), ("", rtol32)):
        for n, ref in zip(names, gold[pre + "grad_norms"]):
            got = grad_of(n)
            if ref < 0:
                assert got is None or float(got.abs().max()) == 0.0, f"{n} should have no gradient"
                continue
            assert got is not None, f"{n}: missing gradient"
            v = got.double().norm().item()
            rt = (5 * rtol if got.numel() == 1 else rtol) * (scale(n) if scale else 1.0)
            # atol: conv/linear biases in front of a BatchNorm have a mathematically zero gradient (rounding noise)
            assert abs(v - ref) <= rt * ref + 1e-6 * total, f"{pre}grad norm {n}: {v} vs {ref}"
        for k in [k for k in gold if k.startswith(pre + "grad:")]:
            name = k[len(pre) + 5:].split("@")[0]
            if k.endswith("@stats"):
                continue
            check_against(gold, pre + "grad:" + name, grad_of(name),
                          (5 * rtol if grad_of(name).numel() == 1 else rtol) * (scale(name) if scale else 1.0))
