"""Generate tests/golden/*.npz from the REAL reference modules (run in the build container only;
/root/reference does not exist on the GPU box).  Usage:  python tests/golden/make_golden.py

The reference is imported with empty stub modules for IPython / cv2 / torchvision (imported
but unused on this path).  No reference source is copied: the fixtures hold only
inputs-by-formula (oracle/procgen.py) and expected outputs.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

for n in ["IPython", "cv2", "torchvision", "torchvision.models", "torchvision.transforms"]:
    sys.modules[n] = types.ModuleType(n)
sys.modules["IPython"].embed = lambda *a, **k: None
sys.modules["torchvision"].models = sys.modules["torchvision.models"]
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.path.insert(0, "/root/reference")

from model import tsrn as ref_tsrn                      # noqa: E402
from model import torch_distortion                      # noqa: E402  (model/__init__.py:4-30)
from loss.image_loss import ImageLoss                   # noqa: E402
from loss.semantic_loss import SemanticLoss             # noqa: E402
from utils import ssim_psnr                             # noqa: E402
from oracle import procgen                              # noqa: E402

MAXS = 4096


def summarize(t: torch.Tensor):
    """(strided sample, [sum, sumsq, numel]) — enough to pin a tensor without storing it."""
    f = t.detach().reshape(-1).double()
    stride = max(1, f.numel() // MAXS)
    return f[::stride][:MAXS].float().numpy(), np.array([f.sum().item(), (f * f).sum().item(), f.numel()], np.float64)


def put(out, name, t, full=False):
    if full:
        out[name] = t.detach().float().numpy()
    else:
        s, st = summarize(t)
        out[name + "@sample"] = s
        out[name + "@stats"] = st


def build(K):
    torch.manual_seed(0)
    m = ref_tsrn.TSRN_TL_TRANS(scale_factor=2, width=128, height=32, STN=True, mask=True,
                               srb_nums=5, hidden_units=32, text_embedding=K)
    m.load_state_dict(procgen.fill_state(m.state_dict(), seed=0), strict=True)
    for mod in m.modules():                             # parity is defined with dropout off
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    return m


def rotate_ref(img, arc):
    """TextSR.torch_rotate_img(img, arc) (interfaces/super_resolution_eng.py:50-82, a method of a class that cannot be
    imported here) through the reference's own importable twin model.torch_distortion (model/__init__.py:4-30):
    with rand_offs = 0.5 its ratios_mul = ratios + 0.5*0.2*2 - 0.2 equals ratios, i.e. the rand_offs=None branch."""
    return torch_distortion(img, arc.to(img.dtype), torch.full_like(arc, 0.5, dtype=img.dtype))


def rot_arcs(B, seed=0, deg=10.0):
    """angle_batch of the TSC term / config 4's pre-deformation: U(-deg, deg) per sample, radians (procedural)."""
    r = np.random.default_rng([B, seed, 77])
    return torch.from_numpy(((r.random(B) * 2 - 1) * deg / 180.0 * np.pi).astype(np.float32))


def case(K, B, train, rot=False):
    """fp32 pass of the reference (keys as is) + fp64 pass of the same modules (keys prefixed 'f64/'): the fp64
    run is the reference algorithm without fp32 rounding noise, which for some gradients (init_factor, PReLU
    slopes, STN head at tiny batch) is as large as 5e-3 relative in the reference's own fp32 run."""
    out = _case(K, B, train, torch.float32, rot)
    if train:
        for k, v in _case(K, B, train, torch.float64, rot).items():
            if k.startswith(("sr", "grad", "image_loss", "loss", "ctrl_points", "tps_out", "tp_map", "query_pos", "pre_sigmoid")):
                out["f64/" + k] = v
    return out


def _case(K, B, train, dtype, rot=False):
    """rot: BASELINE config 4 -- the LR/HR pair is pre-deformed by the reference's rotation (+-10 degrees) before the step, so
    that the STN head / TPS rectification see rotated text."""
    out = {}
    m = build(K).to(dtype)
    hr, lr, tp = [t.to(dtype) for t in procgen.make_inputs(B, K, seed=0)]
    if rot:
        arc = rot_arcs(B)
        out["arc"] = arc.numpy()
        lr, hr = rotate_ref(lr, arc), rotate_ref(hr, arc)
        put(out, "rot_lr", lr, full=(B <= 4))
    inter = {}
    hooks = []

    def hook(name, sel=lambda o: o):
        return lambda mod, i, o: inter.__setitem__(name, sel(o))

    hooks.append(m.stn_head.register_forward_hook(hook("ctrl_points", lambda o: o[1])))
    hooks.append(m.tps.register_forward_hook(hook("tps_out", lambda o: o[0])))
    hooks.append(m.block1.register_forward_hook(hook("block1")))
    hooks.append(m.infoGen.register_forward_hook(hook("tp_map", lambda o: o[0])))
    hooks.append(m.infoGen.transformer.encoder.register_forward_hook(hook("memory")))
    hooks.append(m.infoGen.transformer.gru_encoding.register_forward_hook(
        hook("query_pos", lambda o: o[0].reshape(64, B, 16, 64).permute(2, 0, 1, 3).reshape(1024, B, 64))))
    for i in range(2, 8):
        hooks.append(getattr(m, f"block{i}").register_forward_hook(hook(f"block{i}")))
    hooks.append(m.block8.register_forward_hook(hook("pre_sigmoid")))
    m.train(train)
    sr = m(lr, tp)
    put(out, "sr", sr, full=(B <= 2))
    for k, v in inter.items():
        put(out, k, v, full=(k == "ctrl_points"))
    if train:
        crit = ImageLoss(gradient=True, loss_weight=[1, 1e-4])
        per = crit(sr, hr)
        out["image_loss"] = per.detach().float().numpy()
        loss = per.mean() * 100
        out["loss"] = np.array(loss.item())
        loss.backward()
        names, norms = [], []
        for k, p in m.named_parameters():
            names.append(k)
            norms.append(-1.0 if p.grad is None else p.grad.double().norm().item())
        out["grad_names"] = np.array(names)
        out["grad_norms"] = np.array(norms)
        for k in ["block1.0.weight", "infoGen.fc_in.weight", "stn_head.stn_fc2.bias", "block8.1.weight",
                  "block3.gru1.gru.weight_hh_l0", "block2.bn1.weight", "infoGen.activation.weight",
                  "infoGen.transformer.decoder.norm.weight", "stn_head.stn_convnet.0.0.weight",
                  "infoGen.transformer.encoder.layers.0.self_attn.in_proj_weight"]:
            out["grad:" + k] = dict(m.named_parameters())[k].grad.float().numpy().copy()   # copy: clip_grad_norm_ scales grads in place later
        put(out, "grad:infoGen.transformer.gru_encoding.weight_hh_l0",
            dict(m.named_parameters())["infoGen.transformer.gru_encoding.weight_hh_l0"].grad)
        put(out, "grad:infoGen.init_factor.weight", m.infoGen.init_factor.weight.grad)
        sd = m.state_dict()
        for k in ["block2.bn1.running_mean", "block2.bn1.running_var", "block7.1.running_var",
                  "stn_head.stn_fc1.1.running_mean", "stn_head.stn_convnet.10.1.running_var"]:
            out["bn:" + k] = sd[k].float().numpy().copy()
        total = torch.nn.utils.clip_grad_norm_(m.parameters(), 0.25)
        out["grad_total_norm"] = np.array(float(total))
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.5, 0.999))
        opt.step()
        sd = m.state_dict()
        for k in ["block1.0.bias", "stn_head.stn_fc2.bias", "block8.1.bias", "block4.gru2.gru.bias_hh_l0_reverse"]:
            out["adam:" + k] = sd[k].float().numpy().copy()
    for h in hooks:
        h.remove()
    return out


def losses():
    out = {}
    B = 3
    hr, lr, _ = procgen.make_inputs(B, 37, seed=1)
    g = torch.Generator().manual_seed(5)
    sr = (hr + 0.1 * torch.randn(hr.shape, generator=g)).clamp(0, 1)
    sr2 = (hr + 0.15 * torch.randn(hr.shape, generator=g)).clamp(0, 1)
    out["noise_sr"], out["noise_sr2"] = sr.numpy(), sr2.numpy()
    out["image_loss"] = ImageLoss(gradient=True, loss_weight=[1, 1e-4])(sr, hr).numpy()
    arc = torch.tensor([0.1, -0.17, 0.05])
    out["arc"] = arc.numpy()
    out["rot_hr"] = rotate_ref(hr, arc).numpy()
    out["rot_lr"] = rotate_ref(lr, arc).numpy()
    out["tri_ssim"] = np.array(ssim_psnr.TRI_SSIM()(sr, sr2, hr).item())
    out["ssim"] = np.array(ssim_psnr.SSIM()(sr, hr).item())
    out["psnr"] = np.array(float(ssim_psnr.calculate_psnr(sr, hr)))
    # SemanticLoss (loss/semantic_loss.py:21-38) on two softmaxed [26, B, 37] priors, value and gradient wrt the prediction
    g2 = torch.Generator().manual_seed(9)
    pred = torch.softmax(2.0 * torch.randn(26, B, 37, generator=g2), -1).requires_grad_(True)
    gt = torch.softmax(2.0 * torch.randn(26, B, 37, generator=g2), -1)
    sl = SemanticLoss()(pred, gt)
    sl.backward()
    out["sem_pred"], out["sem_gt"] = pred.detach().numpy(), gt.numpy()
    out["sem_loss"], out["sem_dpred"] = np.array(sl.item()), pred.grad.numpy()
    return out


CASES = [(37, 1, False, False), (37, 2, False, False), (37, 5, False, False), (37, 2, True, False), (37, 5, True, False),
         (2448, 2, True, False),
         (37, 64, True, False),         # BASELINE config 2 at its own size: 64-step RPE chain, well-conditioned train-mode BN
         (37, 4, True, True), (37, 64, True, True)]   # BASELINE config 4: STN/TPS on inputs pre-rotated by +-10 degrees

if __name__ == "__main__":
    torch.set_num_threads(8)
    only = sys.argv[1:]                                  # e.g. `make_golden.py tsrn_K37_B64_train losses` regenerates those only
    for K, B, train, rot in CASES:
        name = f"tsrn_K{K}_B{B}_{'rot_' if rot else ''}{'train' if train else 'eval'}"
        if only and name not in only:
            continue
        o = case(K, B, train, rot)
        fn = os.path.join(HERE, name + ".npz")
        np.savez_compressed(fn, **o)
        print(fn, os.path.getsize(fn) // 1024, "KiB")
    if not only or "losses" in only:
        fn = os.path.join(HERE, "losses.npz")
        np.savez_compressed(fn, **losses())
        print(fn, os.path.getsize(fn) // 1024, "KiB")
