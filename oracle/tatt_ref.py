"""CPU oracle for the TATT super-resolution hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``tatt_amd/`` may import this module:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and there only as the checker / reported baseline.

This is a functional restatement, in stock torch-CPU fp32 ops, of the reference
algorithm (gjh8760/TATT @ 2025-02-24).  Every function cites the reference
file:line it follows.  Parameters are passed as a flat ``dict`` keyed by the
reference's ``state_dict`` names, so one set of tensors can feed
the reference, this oracle and the HIP path.

Parity pin: ``tests/golden/*.npz`` were generated in the build container by
``tests/golden/make_golden.py``, which imports the real reference modules from
``/root/reference`` (with empty stubs for IPython/cv2/torchvision) and runs them
on procedurally generated weights; ``tests/test_oracle_golden.py`` checks this
file against those vectors.  The reference has no tests/golden vectors of its
own for this path.

Dropout: the reference uses torch's RNG stream, which cannot be matched by a
different implementation.  The oracle therefore takes an explicit ``drop``
object (see ``HashDropout``) that produces keep-masks from a counter-based hash
which the HIP kernels implement bit-identically; ``drop=None`` = dropout off.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

# ----------------------------------------------------------------------------
# counter-based dropout (shared definition with tatt_amd/csrc/common.h)
# ----------------------------------------------------------------------------

DROP_SITES = {
    # site ids; the HIP engine uses the same table (tatt_amd/engine.py)
    "pe": 1,
    "enc_attn": 2, "enc_drop1": 3, "enc_ffn": 4, "enc_drop2": 5,
    "dec0_attn": 6, "dec0_drop2": 7, "dec0_ffn": 8, "dec0_drop3": 9,
    "dec1_attn": 10, "dec1_drop2": 11, "dec1_ffn": 12, "dec1_drop3": 13,
}


def hash_u32(seed: int, site: int, idx: np.ndarray) -> np.ndarray:
    """32-bit mix of (seed, site, element index) -> uniform u32 (numpy, wraps mod 2^32)."""
    x = (idx.astype(np.uint64) * np.uint64(0x9E3779B1) + np.uint64((seed * 0x85EBCA77 + site * 0xC2B2AE3D) & 0xFFFFFFFF)) & np.uint64(0xFFFFFFFF)
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x.astype(np.uint64) * np.uint64(0x7FEB352D) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x.astype(np.uint64) * np.uint64(0x846CA68B) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


class HashDropout:
    """keep-mask(site, logical index) = hash >> 8 >= p * 2^24 ; scale 1/(1-p)."""

    def __init__(self, p: float, seed: int):
        self.p = float(p)
        self.seed = int(seed)
        self.thresh = int(self.p * (1 << 24))

    def mask(self, site: str, shape, index: Optional[np.ndarray] = None) -> Tensor:
        n = int(np.prod(shape))
        idx = np.arange(n, dtype=np.uint64) if index is None else index.reshape(-1)
        u = hash_u32(self.seed, DROP_SITES[site], idx) >> np.uint32(8)
        keep = (u >= np.uint32(self.thresh)).astype(np.float32) / np.float32(1.0 - self.p)
        return torch.from_numpy(keep.reshape(shape))


def _drop(x: Tensor, drop: Optional[HashDropout], site: str, index=None) -> Tensor:
    if drop is None or drop.p == 0.0:
        return x
    return x * drop.mask(site, tuple(x.shape), index)


def _drop_seq(x: Tensor, drop: Optional[HashDropout], site: str) -> Tensor:
    """Dropout on a seq-first [L,N,E] tensor; the mask index is (n*L + l)*E + e, i.e. the element's
    offset in the (batch, position)-major row layout the HIP engine stores these tensors in."""
    if drop is None or drop.p == 0.0:
        return x
    L, N, E = x.shape
    l, n, e = np.meshgrid(np.arange(L), np.arange(N), np.arange(E), indexing="ij")
    return _drop(x, drop, site, ((n * L + l) * E + e).astype(np.uint64))


# ----------------------------------------------------------------------------
# small pieces
# ----------------------------------------------------------------------------

def mish(x: Tensor) -> Tensor:
    """model/tsrn.py:1069-1072  x*tanh(softplus(x))."""
    return x * torch.tanh(F.softplus(x))


def batchnorm(P: Params, pre: str, x: Tensor, training: bool, momentum=0.1, eps=1e-5) -> Tensor:
    """nn.BatchNorm2d/1d: batch stats (biased var) in training + running-stat update
    with unbiased var; running stats in eval.  Updates P in place like the module."""
    dims = [0] + list(range(2, x.dim()))
    shape = [1, -1] + [1] * (x.dim() - 2)
    if training:
        mean = x.mean(dims)
        var = x.var(dims, unbiased=False)
        n = x.numel() // x.shape[1]
        with torch.no_grad():
            P[pre + "running_mean"].mul_(1 - momentum).add_(momentum * mean.detach())
            P[pre + "running_var"].mul_(1 - momentum).add_(momentum * var.detach() * (n / max(n - 1, 1)))
            P[pre + "num_batches_tracked"] += 1
    else:
        mean, var = P[pre + "running_mean"], P[pre + "running_var"]
    xh = (x - mean.view(shape)) * torch.rsqrt(var.view(shape) + eps)
    return xh * P[pre + "weight"].view(shape) + P[pre + "bias"].view(shape)


def layernorm(P: Params, pre: str, x: Tensor, eps=1e-5) -> Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * P[pre + "weight"] + P[pre + "bias"]


def gru_dir(x: Tensor, w_ih, w_hh, b_ih, b_hh, reverse: bool) -> Tensor:
    """One direction of nn.GRU, batch_first input [N,T,I] -> [N,T,H].
    Gate order (r,z,n); n = tanh(W_in x + b_in + r*(W_hn h + b_hn)); h' = (1-z) n + z h."""
    N, T, _ = x.shape
    H = w_hh.shape[1]
    gi = x @ w_ih.t() + b_ih                     # [N,T,3H]
    h = x.new_zeros(N, H)
    outs = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        gh = h @ w_hh.t() + b_hh
        r = torch.sigmoid(gi[:, t, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, t, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, t, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        outs[t] = h
    return torch.stack(outs, 1)


def bigru(P: Params, pre: str, x: Tensor, fused: bool = False) -> Tensor:
    """nn.GRU(bidirectional=True, batch_first=True), one layer, h0 = 0."""
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    wf = [P[pre + n] for n in names]
    wb = [P[pre + n + "_reverse"] for n in names]
    if fused:  # same math through torch's native CPU GRU (used only to time the CPU baseline)
        h0 = x.new_zeros(2, x.shape[0], wf[1].shape[1])
        out, _ = torch._VF.gru(x, h0, wf + wb, True, 1, 0.0, False, True, True)
        return out
    return torch.cat([gru_dir(x, *wf, reverse=False), gru_dir(x, *wb, reverse=True)], -1)


def mha(P: Params, pre: str, q_in: Tensor, k_in: Tensor, v_in: Tensor, nhead: int,
        drop: Optional[HashDropout], site: str):
    """nn.MultiheadAttention (seq-first [L,N,E]), packed in-proj, q scaled by 1/sqrt(dh),
    dropout on the softmax weights, head-averaged weights returned."""
    L, N, E = q_in.shape
    S = k_in.shape[0]
    dh = E // nhead
    W, b = P[pre + "in_proj_weight"], P[pre + "in_proj_bias"]
    q = q_in @ W[:E].t() + b[:E]
    k = k_in @ W[E:2 * E].t() + b[E:2 * E]
    v = v_in @ W[2 * E:].t() + b[2 * E:]
    q = q.reshape(L, N, nhead, dh).permute(1, 2, 0, 3) * (1.0 / math.sqrt(dh))   # [N,h,L,dh]
    k = k.reshape(S, N, nhead, dh).permute(1, 2, 0, 3)
    v = v.reshape(S, N, nhead, dh).permute(1, 2, 0, 3)
    att = torch.softmax(q @ k.transpose(-1, -2), -1)                               # [N,h,L,S]
    att_d = _drop(att, drop, site)
    ctx = (att_d @ v).permute(2, 0, 1, 3).reshape(L, N, E)
    out = ctx @ P[pre + "out_proj.weight"].t() + P[pre + "out_proj.bias"]
    # torch averages the (post-dropout) weights over heads
    return out, att_d.mean(1)


# ----------------------------------------------------------------------------
# STN head + TPS
# ----------------------------------------------------------------------------

def stn_head(P: Params, x: Tensor, training: bool, pre="stn_head."):
    """model/stn_head.py:92-106 (net :32-53)."""
    pools = {0: (2, 2), 2: (2, 2), 4: (2, 2), 6: (2, 2), 8: (1, 2), 10: None}
    for i in (0, 2, 4, 6, 8, 10):
        x = F.conv2d(x, P[f"{pre}stn_convnet.{i}.0.weight"], P[f"{pre}stn_convnet.{i}.0.bias"], padding=1)
        x = torch.relu(batchnorm(P, f"{pre}stn_convnet.{i}.1.", x, training))
        if pools[i] is not None:
            x = F.max_pool2d(x, pools[i], pools[i])
    x = x.reshape(x.shape[0], -1)
    feat = x @ P[pre + "stn_fc1.0.weight"].t() + P[pre + "stn_fc1.0.bias"]
    feat = torch.relu(batchnorm(P, pre + "stn_fc1.1.", feat, training))
    ctrl = (0.1 * feat) @ P[pre + "stn_fc2.weight"].t() + P[pre + "stn_fc2.bias"]
    return feat, ctrl.view(-1, 20, 2)


def tps_buffers(height=16, width=64, n_ctrl=20, margins=(0.05, 0.05)):
    """model/tps_spatial_transformer.py:22-34, :38-50, :62-95 — precomputed TPS matrices."""
    side = n_ctrl // 2
    xs = np.linspace(margins[0], 1.0 - margins[0], side)
    top = np.stack([xs, np.full(side, margins[1])], 1)
    bot = np.stack([xs, np.full(side, 1.0 - margins[1])], 1)
    ctrl = torch.tensor(np.concatenate([top, bot], 0), dtype=torch.float32)

    def partial_repr(a, b):
        d = a.view(-1, 1, 2) - b.view(1, -1, 2)
        d2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]
        r = 0.5 * d2 * torch.log(d2)
        return torch.where(torch.isnan(r), torch.zeros_like(r), r)

    N = n_ctrl
    fk = torch.zeros(N + 3, N + 3)
    fk[:N, :N] = partial_repr(ctrl, ctrl)
    fk[:N, -3] = 1
    fk[-3, :N] = 1
    fk[:N, -2:] = ctrl
    fk[-2:, :N] = ctrl.t()
    inv = torch.inverse(fk)
    ys, xs_ = torch.meshgrid(torch.arange(height, dtype=torch.float32), torch.arange(width, dtype=torch.float32), indexing="ij")
    coord = torch.stack([xs_.reshape(-1) / (width - 1), ys.reshape(-1) / (height - 1)], 1)
    rep = torch.cat([partial_repr(coord, ctrl), torch.ones(height * width, 1), coord], 1)
    return {"inverse_kernel": inv, "padding_matrix": torch.zeros(3, 2),
            "target_coordinate_repr": rep, "target_control_points": ctrl}


def grid_sample_bilinear(x: Tensor, grid: Tensor) -> Tensor:
    """F.grid_sample defaults used by the reference (bilinear, zeros, align_corners=False),
    restated explicitly: pixel = ((g+1)*size-1)/2, 4 taps, out-of-range taps contribute 0."""
    B, C, H, W = x.shape
    gx = ((grid[..., 0] + 1) * W - 1) * 0.5
    gy = ((grid[..., 1] + 1) * H - 1) * 0.5
    x0 = torch.floor(gx); y0 = torch.floor(gy)
    wx1 = gx - x0; wy1 = gy - y0
    out = 0
    flat = x.reshape(B, C, H * W)
    for dy, wy in ((0, 1 - wy1), (1, wy1)):
        for dx, wx in ((0, 1 - wx1), (1, wx1)):
            xi = x0 + dx; yi = y0 + dy
            ok = ((xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)).to(x.dtype)
            idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).long().view(B, 1, -1).expand(B, C, -1)
            v = flat.gather(2, idx).view(B, C, *grid.shape[1:3])
            out = out + v * (wx * wy * ok).unsqueeze(1)
    return out


def tps_forward(P: Params, x: Tensor, ctrl: Tensor, pre="tps."):
    """model/tps_spatial_transformer.py:97-112."""
    B = ctrl.shape[0]
    Y = torch.cat([ctrl, P[pre + "padding_matrix"].expand(B, 3, 2)], 1)
    mapping = P[pre + "inverse_kernel"] @ Y
    src = P[pre + "target_coordinate_repr"] @ mapping
    H, W = x.shape[-2:]
    grid = src.view(-1, H, W, 2).clamp(0, 1) * 2.0 - 1.0
    return grid_sample_bilinear(x, grid), src


# ----------------------------------------------------------------------------
# TP interpreter (TPInterpreter + InfoTransformer)
# ----------------------------------------------------------------------------

def positional_encoding(L: int, d: int) -> Tensor:
    """model/transformer_v2.py:30-35."""
    pe = torch.zeros(L, d)
    pos = torch.arange(0, L).unsqueeze(1).float()
    div = torch.exp(torch.arange(0, d, 2).float() * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def rpe_query_pos(P: Params, B: int, H=16, W=64, d=64, pre="infoGen.", fused=False) -> Tensor:
    """Recurrent positional encoding, model/transformer_v2.py:201,:215-221.
    The GRU is batch_first but is fed [W, B, H*d]: it recurs over the BATCH axis
    (SURVEY §8a a7).  Returns query_pos [H*W, B, d]."""
    q = P[pre + "init_factor.weight"].unsqueeze(1).repeat(1, B, 1)           # [HW,B,d]
    q = q.reshape(H, W, B, d).permute(1, 2, 0, 3).reshape(W, B, H * d)
    q = bigru(P, pre + "transformer.gru_encoding.", q, fused=fused)            # seq axis = B
    return q.reshape(W, B, H, d).permute(2, 0, 1, 3).reshape(H * W, B, d)


def encoder_layer(P: Params, pre: str, src: Tensor, pos: Tensor, drop):
    """TransformerEncoder.forward :256-279 (input doubled at :273) + forward_post :469-483."""
    src = src + src
    qk = src + pos
    a, _ = mha(P, pre + "self_attn.", qk, qk, src, 4, drop, "enc_attn")
    src = layernorm(P, pre + "norm1.", src + _drop_seq(a, drop, "enc_drop1"))
    h = torch.relu(src @ P[pre + "linear1.weight"].t() + P[pre + "linear1.bias"])
    h = _drop_seq(h, drop, "enc_ffn") @ P[pre + "linear2.weight"].t() + P[pre + "linear2.bias"]
    return layernorm(P, pre + "norm2.", src + _drop_seq(h, drop, "enc_drop2"))


def decoder_layer(P: Params, pre: str, li: int, tgt: Tensor, memory: Tensor, pos: Tensor, query_pos: Tensor, drop):
    """TransformerDecoderLayer_TP.forward_post :805-832 (self_attn/norm1 never run)."""
    a, w = mha(P, pre + "multihead_attn.", tgt + query_pos, memory + pos, memory, 4, drop, f"dec{li}_attn")
    tgt = layernorm(P, pre + "norm2.", tgt + _drop_seq(a, drop, f"dec{li}_drop2"))
    h = torch.relu(tgt @ P[pre + "linear1.weight"].t() + P[pre + "linear1.bias"])
    h = _drop_seq(h, drop, f"dec{li}_ffn") @ P[pre + "linear2.weight"].t() + P[pre + "linear2.bias"]
    return layernorm(P, pre + "norm3.", tgt + _drop_seq(h, drop, f"dec{li}_drop3")), w


def tp_interpreter(P: Params, feat: Tensor, tp: Tensor, drop=None, pre="infoGen.", inter=None, fused=False):
    """TPInterpreter.forward model/tsrn.py:194-226 + InfoTransformer.forward transformer_v2.py:198-244.
    Dropout index convention: masks are indexed by the logical element order of the tensor
    in the layouts used here ([L,N,E] seq-first; attention [N,h,L,S]; pe [N,L,C])."""
    B, C, H, W = feat.shape
    x_im = feat.reshape(B, C, H * W).permute(2, 0, 1)                         # [HW,B,64]
    t = tp.permute(0, 3, 1, 2).squeeze(-1)                                    # [B,26,K]
    t = F.prelu(t @ P[pre + "fc_in.weight"].t() + P[pre + "fc_in.bias"], P[pre + "activation.weight"])
    L = t.shape[1]
    pos = _drop(positional_encoding(L, C).unsqueeze(0).expand(B, L, C), drop, "pe").permute(1, 0, 2)
    t = t.permute(1, 0, 2)                                                    # [26,B,64]
    qp = rpe_query_pos(P, B, H, W, C, pre, fused=fused)
    mem = encoder_layer(P, pre + "transformer.encoder.layers.0.", t, pos, drop)
    out = x_im
    hs = []
    w = None
    for li in range(2):
        out, w = decoder_layer(P, f"{pre}transformer.decoder.layers.{li}.", li, out, mem, pos, qp, drop)
        hs.append(layernorm(P, pre + "transformer.decoder.norm.", out))       # :380
    tp_map = torch.stack(hs).mean(0).permute(1, 2, 0).reshape(B, C, H, W)      # tsrn.py:221-224
    if inter is not None:
        inter.update(query_pos=qp, memory=mem, tp_map=tp_map, attn=w)
    return tp_map, w


# ----------------------------------------------------------------------------
# SR trunk
# ----------------------------------------------------------------------------

def gru_block(P: Params, pre: str, x: Tensor, fused=False) -> Tensor:
    """GruBlock.forward model/tsrn.py:1082-1092."""
    x = F.conv2d(x, P[pre + "conv1.weight"], P[pre + "conv1.bias"])
    B, C, H, W = x.shape
    s = x.permute(0, 2, 3, 1).reshape(B * H, W, C)
    o = bigru(P, pre + "gru.", s, fused=fused)
    return o.view(B, H, W, C).permute(0, 3, 1, 2)


def srb(P: Params, pre: str, x: Tensor, tp_map: Tensor, training: bool, fused=False) -> Tensor:
    """RecurrentResidualBlockTL.forward model/tsrn.py:883-901."""
    r = F.conv2d(x, P[pre + "conv1.weight"], P[pre + "conv1.bias"], padding=1)
    r = mish(batchnorm(P, pre + "bn1.", r, training))
    r = F.conv2d(r, P[pre + "conv2.weight"], P[pre + "conv2.bias"], padding=1)
    r = batchnorm(P, pre + "bn2.", r, training)
    cat = torch.cat([r, tp_map], 1)
    r = gru_block(P, pre + "gru1.", cat.transpose(-1, -2), fused).transpose(-1, -2)
    return gru_block(P, pre + "gru2.", x + r, fused)


def tsrn_forward(P: Params, x: Tensor, text_emb: Optional[Tensor] = None, training: bool = False,
                 stn: bool = True, srb_nums: int = 5, drop: Optional[HashDropout] = None,
                 inter: Optional[dict] = None, fused_rnn: bool = False) -> Tensor:
    """TSRN_TL_TRANS.forward model/tsrn.py:648-684."""
    if stn and training:
        _, ctrl = stn_head(P, x, training)
        x, _ = tps_forward(P, x, ctrl)
        if inter is not None:
            inter.update(ctrl_points=ctrl, tps_out=x)
    b1 = F.prelu(F.conv2d(x, P["block1.0.weight"], P["block1.0.bias"], padding=4), P["block1.1.weight"])
    if text_emb is None:
        K = P["infoGen.fc_in.weight"].shape[1]
        text_emb = torch.zeros(x.shape[0], K, 1, 26)
    tp_map, _ = tp_interpreter(P, b1, text_emb, drop if training else None, inter=inter, fused=fused_rnn)
    blk = b1
    if inter is not None:
        inter["block1"] = b1
    for i in range(srb_nums):
        blk = srb(P, f"block{i + 2}.", blk, tp_map, training, fused_rnn)
        if inter is not None:
            inter[f"block{i + 2}"] = blk
    k = srb_nums + 2
    b7 = batchnorm(P, f"block{k}.1.", F.conv2d(blk, P[f"block{k}.0.weight"], P[f"block{k}.0.bias"], padding=1), training)
    k += 1
    u = F.conv2d(b1 + b7, P[f"block{k}.0.conv.weight"], P[f"block{k}.0.conv.bias"], padding=1)
    u = mish(F.pixel_shuffle(u, 2))                                            # UpsampleBLock :1040-1044
    pre_sig = F.conv2d(u, P[f"block{k}.1.weight"], P[f"block{k}.1.bias"], padding=4)
    if inter is not None:
        inter.update(block7=b7, pre_sigmoid=pre_sig)
    return torch.sigmoid(pre_sig)


# ----------------------------------------------------------------------------
# losses, TSC deformation, metrics, optimiser
# ----------------------------------------------------------------------------

def gradient_map(x: Tensor) -> Tensor:
    """loss/image_loss.py:49-56."""
    r = F.pad(x, (0, 1, 0, 0))[:, :, :, 1:]
    l = F.pad(x, (1, 0, 0, 0))[:, :, :, :-1]
    t = F.pad(x, (0, 0, 1, 0))[:, :, :-1, :]
    b = F.pad(x, (0, 0, 0, 1))[:, :, 1:, :]
    return torch.sqrt(((r - l) * 0.5) ** 2 + ((t - b) * 0.5) ** 2 + 1e-6)


def image_loss(sr: Tensor, hr: Tensor, gradient=True, loss_weight=(1.0, 1e-4)) -> Tensor:
    """ImageLoss.forward loss/image_loss.py:19-32 -> per-sample loss [B]."""
    mse = ((sr - hr) ** 2).mean((1, 2, 3))
    if not gradient:
        return loss_weight[0] * mse
    g = (gradient_map(sr[:, :3]) - gradient_map(hr[:, :3])).abs().mean((1, 2, 3))
    return loss_weight[0] * mse + loss_weight[1] * g


def rotate(img: Tensor, arc: Tensor) -> Tensor:
    """torch_rotate_img with rand_offs=None, interfaces/super_resolution_eng.py:50-82:
    theta = [[cos, sin*r, 0], [-sin/r, cos, 0]], r = H/W; affine_grid + grid_sample defaults."""
    N, C, H, W = img.shape
    r = H / float(W)
    arc = arc.to(img.dtype)
    c, s = torch.cos(arc), torch.sin(arc)
    xs = (2 * torch.arange(W, dtype=img.dtype) + 1) / W - 1                  # align_corners=False base grid
    ys = (2 * torch.arange(H, dtype=img.dtype) + 1) / H - 1
    gx = c.view(N, 1, 1) * xs.view(1, 1, W) + (s * r).view(N, 1, 1) * ys.view(1, H, 1)
    gy = (-s / r).view(N, 1, 1) * xs.view(1, 1, W) + c.view(N, 1, 1) * ys.view(1, H, 1)
    return grid_sample_bilinear(img, torch.stack([gx, gy], -1))


def gaussian_window(size=11, sigma=1.5) -> Tensor:
    """utils/ssim_psnr.py:28-37."""
    g = torch.tensor([math.exp(-(x - size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(size)])
    g = g / g.sum()
    return (g.unsqueeze(1) @ g.unsqueeze(0)).float()


def _blur(x: Tensor, win: Tensor) -> Tensor:
    C = x.shape[1]
    win = win.to(x.dtype)           # the fp32 window values, whatever precision the referee run uses
    return F.conv2d(x, win.expand(C, 1, *win.shape).contiguous(), padding=win.shape[0] // 2, groups=C)


def tri_ssim(a: Tensor, b: Tensor, c: Tensor) -> Tensor:
    """TRI_SSIM / _tri_ssim utils/ssim_psnr.py:99-128 (all channels, global mean)."""
    w = gaussian_window()
    ma, mb, mc = _blur(a, w), _blur(b, w), _blur(c, w)
    sa = _blur(a * a, w) - ma * ma
    sb = _blur(b * b, w) - mb * mb
    sc = _blur(c * c, w) - mc * mc
    sab = _blur(a * b, w) - ma * mb
    sbc = _blur(b * c, w) - mb * mc
    sca = _blur(c * a, w) - mc * ma
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((ma * mb + mb * mc + mc * ma + C1) * (sab + sbc + sca + C2)) / \
        ((ma * ma + mb * mb + mc * mc + C1) * (sa + sb + sc + C2))
    return m.mean()


def ssim(a: Tensor, b: Tensor) -> Tensor:
    """SSIM.forward / _ssim utils/ssim_psnr.py:76-96,:200-225 (channels 0-2, size_average)."""
    a, b = a[:, :3], b[:, :3]
    w = gaussian_window()
    ma, mb = _blur(a, w), _blur(b, w)
    sa = _blur(a * a, w) - ma * ma
    sb = _blur(b * b, w) - mb * mb
    sab = _blur(a * b, w) - ma * mb
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * ma * mb + C1) * (2 * sab + C2)) / ((ma * ma + mb * mb + C1) * (sa + sb + C2))).mean()


def psnr(a: Tensor, b: Tensor) -> Tensor:
    """calculate_psnr utils/ssim_psnr.py:9-15."""
    mse = ((a[:, :3] * 255 - b[:, :3] * 255) ** 2).mean()
    return 20 * torch.log10(255.0 / torch.sqrt(mse))


def clip_and_adam(params, grads, state, step: int, lr=1e-3, betas=(0.5, 0.999), eps=1e-8, max_norm=0.25):
    """clip_grad_norm_(params, 0.25) + Adam step; interfaces/super_resolution_eng.py:277-279,
    interfaces/base.py:232-233.  params/grads: lists (None grads skipped). Returns total norm."""
    gs = [g for g in grads if g is not None]
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in gs)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    b1, b2 = betas
    for i, (p, g) in enumerate(zip(params, grads)):
        if g is None:
            continue
        g = g * coef
        m, v = state.setdefault(i, (torch.zeros_like(p), torch.zeros_like(p)))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(1 - b2 ** step)).add_(eps)
        p.data.addcdiv_(m, denom, value=-lr / (1 - b1 ** step))
    return total


def loss_stablizing(loss_set: Tensor, keep_proportion: float = 0.7) -> Tensor:
    """TextSR.loss_stablizing interfaces/super_resolution_eng.py:38-48, quirk included: everything from the ORIGINAL
    position `sorted_ind[int(keep*B)]` onward is zeroed (a data index used as a slice start, not a rank)."""
    _, sorted_ind = torch.sort(loss_set)
    start = int(sorted_ind[int(keep_proportion * loss_set.shape[0])])
    mask = torch.ones_like(loss_set)
    mask[start:] = 0.0
    return loss_set * mask


def tsc_losses(P: Params, lr_img: Tensor, hr_img: Tensor, tp: Tensor, arc: Tensor, drop: Optional[HashDropout] = None,
               drop2: Optional[HashDropout] = None, training_stablize: bool = False, fused_rnn: bool = False):
    """L_SR + L_TSC of the reference step, interfaces/super_resolution_eng.py:233-262:
    loss_img = ImageLoss(sr, hr).mean()*100 + (1 - TRI_SSIM(D(F(Y)), F(D(Y)), D(X)))*10, with a second SR forward on
    the rotated LR input (BatchNorm statistics update twice, fresh dropout masks = `drop2`)."""
    sr = tsrn_forward(P, lr_img, tp, training=True, drop=drop, fused_rnn=fused_rnn)
    per = image_loss(sr, hr_img)
    if training_stablize:
        per = loss_stablizing(per)
    loss_img = per.mean() * 100
    deformed_sr = rotate(sr, arc)
    deformed_lr = rotate(lr_img, arc)
    deformed_first_sr = tsrn_forward(P, deformed_lr, tp, training=True, drop=drop2, fused_rnn=fused_rnn)
    deformed_hr = rotate(hr_img, arc)
    loss_tssim = (1 - tri_ssim(deformed_sr, deformed_first_sr, deformed_hr)) * 10.0
    return loss_img, loss_tssim, sr, deformed_first_sr


def train_step_config3(P: Params, lr_img: Tensor, hr_img: Tensor, tp: Tensor, arc: Tensor, state: dict, step: int,
                       drop: Optional[HashDropout] = None, drop2: Optional[HashDropout] = None,
                       training_stablize: bool = False, fused_rnn: bool = False):
    """BASELINE config 3 step in the arithmetic type of P: config 2 plus the TSC term."""
    names = [k for k, v in P.items() if v.is_floating_point() and v.requires_grad]
    for k in names:
        P[k].grad = None
    loss_img, loss_tssim, sr, sr2 = tsc_losses(P, lr_img, hr_img, tp, arc, drop, drop2, training_stablize, fused_rnn)
    (loss_img + loss_tssim).backward()
    params = [P[k] for k in names]
    with torch.no_grad():
        total = clip_and_adam(params, [p.grad for p in params], state, step)
    return loss_img.detach(), loss_tssim.detach(), total, sr.detach(), sr2.detach()


def train_step_config2(P: Params, lr_img: Tensor, hr_img: Tensor, tp: Tensor, state: dict, step: int,
                       drop: Optional[HashDropout] = None, fused_rnn: bool = False):
    """BASELINE config 2 step: forward (train mode, STN on), ImageLoss.mean()*100,
    backward, clip 0.25, Adam(1e-3, b1 .5).  P holds leaf tensors; float params get grads."""
    names = [k for k, v in P.items() if v.is_floating_point() and v.requires_grad]
    for k in names:
        P[k].grad = None
    sr = tsrn_forward(P, lr_img, tp, training=True, drop=drop, fused_rnn=fused_rnn)
    loss = image_loss(sr, hr_img).mean() * 100
    loss.backward()
    params = [P[k] for k in names]
    with torch.no_grad():
        total = clip_and_adam(params, [p.grad for p in params], state, step)
    return loss.detach(), total, sr.detach()
