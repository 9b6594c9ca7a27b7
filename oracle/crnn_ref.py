"""CPU oracle of the CRNN text-prior generator.  TEST INFRASTRUCTURE ONLY: imported by tests/,
never by the product package.  Own restatement of

  * parse_crnn_data            interfaces/base.py:441-459   (bicubic resize to 32x100, gray mix)
  * CRNN.forward               model/crnn/crnn.py:95-119    (7 convs, 3 BatchNorms, 4 max-pools, 2 BidirectionalLSTM)
  * BidirectionalLSTM.forward  model/crnn/crnn.py:13-41     (nn.LSTM + `embedding` Linear evaluated in float64)

pinned by tests/golden/crnn_*.npz, which tests/golden/make_golden_crnn.py generated from the real reference modules
(tests/test_oracle_golden.py::test_crnn_oracle_vs_reference_golden).
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


def _cubic_matrix(n_in: int, n_out: int, dtype) -> Tensor:
    """[n_out, n_in] interpolation matrix of torch's upsample_bicubic2d (align_corners=False, A=-0.75, border clamp)."""
    A = -0.75
    M = torch.zeros(n_out, n_in, dtype=dtype)
    scale = torch.tensor(n_in / n_out, dtype=torch.float32).item() if dtype == torch.float32 else n_in / n_out
    for o in range(n_out):
        r = scale * (o + 0.5) - 0.5
        i0 = math.floor(r)
        t = r - i0
        w = [((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A,
             ((A + 2) * t - (A + 3)) * t * t + 1,
             ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1,
             ((A * (2 - t) - 5 * A) * (2 - t) + 8 * A) * (2 - t) - 4 * A]
        for q in range(4):
            M[o, min(max(i0 - 1 + q, 0), n_in - 1)] += w[q]
    return M


def parse_crnn_data(imgs: Tensor, out_hw=(32, 100)) -> Tensor:
    """interfaces/base.py:441-459 with ratio_keep=False: imgs [B,>=3,H,W] -> gray [B,1,32,100]."""
    x = imgs[:, :3]
    My = _cubic_matrix(x.shape[2], out_hw[0], x.dtype)
    Mx = _cubic_matrix(x.shape[3], out_hw[1], x.dtype)
    x = torch.einsum("oh,bchw,pw->bcop", My, x, Mx)
    return 0.299 * x[:, 0:1] + 0.587 * x[:, 1:2] + 0.114 * x[:, 2:3]


def _bn(P: Params, pre: str, x: Tensor, training: bool) -> Tensor:
    if training:
        mean, var = x.mean((0, 2, 3)), x.var((0, 2, 3), unbiased=False)
        n = x.numel() / x.shape[1]
        with torch.no_grad():
            P[pre + "running_mean"].mul_(0.9).add_(0.1 * mean.detach().to(P[pre + "running_mean"].dtype))
            P[pre + "running_var"].mul_(0.9).add_(0.1 * (var.detach() * n / (n - 1)).to(P[pre + "running_var"].dtype))
            P[pre + "num_batches_tracked"] += 1
    else:
        mean, var = P[pre + "running_mean"].to(x.dtype), P[pre + "running_var"].to(x.dtype)
    xh = (x - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + 1e-5)
    return xh * P[pre + "weight"].view(1, -1, 1, 1) + P[pre + "bias"].view(1, -1, 1, 1)


def lstm_dir(x: Tensor, w_ih, w_hh, b_ih, b_hh, reverse: bool) -> Tensor:
    """One direction of nn.LSTM: x [T,B,I] -> [T,B,H]; gates (i, f, g, o)."""
    T, B, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    gi = x @ w_ih.t() + b_ih
    out = [None] * T
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        g = gi[t] + h @ w_hh.t() + b_hh
        i, f, gg, o = g.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        out[t] = h
    return torch.stack(out, 0)


def bidirectional_lstm(P: Params, pre: str, x: Tensor) -> Tensor:
    """BidirectionalLSTM.forward model/crnn/crnn.py:13-41: LSTM, then the Linear in float64, result cast to float32."""
    r = pre + "rnn."
    fw = lstm_dir(x, P[r + "weight_ih_l0"], P[r + "weight_hh_l0"], P[r + "bias_ih_l0"], P[r + "bias_hh_l0"], False)
    bw = lstm_dir(x, P[r + "weight_ih_l0_reverse"], P[r + "weight_hh_l0_reverse"], P[r + "bias_ih_l0_reverse"],
                  P[r + "bias_hh_l0_reverse"], True)
    rec = torch.cat([fw, bw], 2)
    T, B, Hh = rec.shape
    out = rec.reshape(T * B, Hh).double() @ P[pre + "embedding.weight"].double().t() + P[pre + "embedding.bias"].double()
    return out.to(x.dtype).view(T, B, -1)


def crnn_forward(P: Params, gray: Tensor, training: bool = False, inter: dict | None = None) -> Tensor:
    """CRNN.forward model/crnn/crnn.py:95-119: gray [B,1,32,W] -> logits [W/4+1, B, nclass]."""
    c = lambda i, x, p=1: F.conv2d(x, P[f"cnn.conv{i}.weight"], P[f"cnn.conv{i}.bias"], padding=p)
    x = F.max_pool2d(F.relu(c(0, gray)), 2, 2)
    x = F.max_pool2d(F.relu(c(1, x)), 2, 2)
    x = F.relu(_bn(P, "cnn.batchnorm2.", c(2, x), training))
    x = F.max_pool2d(F.relu(c(3, x)), (2, 2), (2, 1), (0, 1))
    x = F.relu(_bn(P, "cnn.batchnorm4.", c(4, x), training))
    x = F.max_pool2d(F.relu(c(5, x)), (2, 2), (2, 1), (0, 1))
    x = F.relu(_bn(P, "cnn.batchnorm6.", c(6, x, 0), training))
    assert x.shape[2] == 1, "the height of conv must be 1"
    seq = x.squeeze(2).permute(2, 0, 1)
    mid = bidirectional_lstm(P, "rnn.0.", seq)
    out = bidirectional_lstm(P, "rnn.1.", mid)
    if inter is not None:
        inter.update(cnn=x, rnn0=mid)
    return out


def semantic_loss(pred: Tensor, gt: Tensor) -> Tensor:
    """SemanticLoss.forward loss/semantic_loss.py:21-38 (lambda1 = lambda2 = 1)."""
    return (gt - pred).abs().mean() + torch.nn.KLDivLoss()(torch.log(pred + 1e-20), gt + 1e-20)


def full_train_step(P_sr: Params, P_t: Params, P_s: Params, lr_img: Tensor, hr_img: Tensor, st_sr: dict, st_s: dict, step: int,
                    arc: Tensor | None = None, lr_tpg: float = 1e-4):
    """One iteration of TextSR.train (interfaces/super_resolution_eng.py:214-279) with dropout off:
    teacher / student priors, SR forward on the detached student prior, L_SR (+ L_TSC when `arc` is given) + L_TP*100,
    backward, clip_grad_norm_(SR params, 0.25), Adam(SR lr 1e-3, student lr 1e-4, betas (0.5, 0.999))."""
    from oracle import tatt_ref as R
    for P in (P_sr, P_s):
        for v in P.values():
            if v.is_floating_point() and v.requires_grad:
                v.grad = None
    with torch.no_grad():
        label_vecs_hr = torch.softmax(crnn_forward(P_t, parse_crnn_data(hr_img), training=False), -1)
    logits = crnn_forward(P_s, parse_crnn_data(lr_img), training=True)
    label_vecs = torch.softmax(logits, -1)
    prior = label_vecs.permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2).detach()
    loss_tp = semantic_loss(label_vecs, label_vecs_hr) * 100
    if arc is None:
        sr = R.tsrn_forward(P_sr, lr_img, prior, training=True)
        loss_img, loss_tssim = R.image_loss(sr, hr_img).mean() * 100, torch.zeros(())
    else:
        loss_img, loss_tssim, sr, _ = R.tsc_losses(P_sr, lr_img, hr_img, prior, arc)
    (loss_img + loss_tssim + loss_tp).backward()
    sr_params = [v for v in P_sr.values() if v.is_floating_point() and v.requires_grad]
    s_params = [v for v in P_s.values() if v.is_floating_point() and v.requires_grad]
    with torch.no_grad():
        total = R.clip_and_adam(sr_params, [p.grad for p in sr_params], st_sr, step)
        R.clip_and_adam(s_params, [p.grad for p in s_params], st_s, step, lr=lr_tpg, max_norm=float("inf"))
    return loss_img.detach(), loss_tssim.detach(), loss_tp.detach(), total, sr.detach()


ALPHABET = "-0123456789abcdefghijklmnopqrstuvwxyz"


def get_string_crnn(outputs: Tensor, alphabet: str = ALPHABET) -> list:
    """utils/metrics.py:128-146: per-step argmax, merge runs of equal characters, a blank (class 0) ends a run and is dropped.
    Pinned by tests/golden/ctc_strings.json (made from the reference function)."""
    res = []
    for idx in outputs.permute(1, 0, 2).argmax(-1).tolist():
        s, last = "", None
        for i in idx:
            ch = alphabet[i]
            if i == 0:
                last = None
            elif ch != last:
                s += ch
                last = ch
        res.append(s)
    return res


def str_filt(s: str, voc: str) -> str:
    """utils/util.py:14-42 for a vocabulary given as a string / list of allowed characters."""
    return "".join(c for c in s.lower() if c in voc)


def eval_batches(P_sr: Params, P_stu: Params, P_rec: Params, batches) -> dict:
    """TextSR.eval (interfaces/super_resolution_eng.py:375-552, test_model == 'crnn') on stock torch: per-batch PSNR / SSIM of
    SR and bicubic LR vs HR (channels 0..2), greedy strings of LR / HR / SR, accuracies after str_filt('lower')."""
    from oracle import tatt_ref as R
    voc = "0123456789abcdefghijklmnopqrstuvwxyz"
    md = {"psnr": [], "ssim": [], "psnr_lr": [], "ssim_lr": [], "strings_sr": [], "strings_lr": [], "strings_hr": []}
    n = {"sr": 0, "lr": 0, "hr": 0}
    tot = 0
    with torch.no_grad():
        for hr, lr, labels in batches:
            prior = text_prior(crnn_forward(P_stu, parse_crnn_data(lr), training=False))
            sr = R.tsrn_forward(P_sr, lr, prior, training=False)
            for tag, img in (("lr", lr), ("hr", hr), ("sr", sr)):
                strs = get_string_crnn(crnn_forward(P_rec, parse_crnn_data(img), training=False))
                md["strings_" + tag] += strs
                n[tag] += sum(str_filt(s, voc) == str_filt(l, voc) for s, l in zip(strs, labels))
            up = F.interpolate(lr, sr.shape[-2:], mode="bicubic")
            md["psnr"].append(R.psnr(sr[:, :3], hr[:, :3]))
            md["ssim"].append(R.ssim(sr[:, :3], hr[:, :3]))
            md["psnr_lr"].append(R.psnr(up[:, :3], hr[:, :3]))
            md["ssim_lr"].append(R.ssim(up[:, :3], hr[:, :3]))
            tot += lr.shape[0]
    md["accuracy"], md["accuracy_lr"], md["accuracy_hr"] = (round(n[k] / tot, 4) for k in ("sr", "lr", "hr"))
    md["psnr_avg"] = round(float(sum(md["psnr"]) / (len(md["psnr"]) + 1e-10)), 6)
    md["ssim_avg"] = round(float(sum(md["ssim"]) / (len(md["psnr"]) + 1e-10)), 6)
    return md


def text_prior(logits: Tensor) -> Tensor:
    """label_vecs_final of the trainer (interfaces/super_resolution_eng.py:227-231): softmax over classes, [T,B,K] ->
    the [B,K,1,T] view that TSRN_TL_TRANS.forward takes."""
    return torch.softmax(logits, -1).permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2)
