"""Procedural (name-hashed) parameters and synthetic inputs shared by the golden-fixture
generator, the parity tests, smoke() and bench.py's cpu_baseline.  TEST INFRASTRUCTURE:
not imported by the product package.

Values depend only on (tensor name, shape, seed), so 30 MB of weights are regenerated
instead of stored.  Unlike the reference's init (stn_fc2 weight = 0, BN = identity) every
tensor is non-trivial so that every gradient path is exercised.
"""
from __future__ import annotations

import zlib
from typing import Dict, Mapping

import numpy as np
import torch


def _rng(name: str, seed: int) -> np.random.Generator:
    return np.random.default_rng([zlib.crc32(name.encode()), seed])


def tensor_for(name: str, shape, seed: int = 0, ref: torch.Tensor | None = None) -> torch.Tensor:
    shape = tuple(shape)
    r = _rng(name, seed)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if name.startswith("tps.") or name.endswith("pe.pe"):
        assert ref is not None, name          # deterministic buffers: keep the module's own value
        return ref.clone()
    if leaf == "running_mean":
        v = 0.1 * r.standard_normal(shape)
    elif leaf == "running_var":
        v = 1.0 + 0.3 * r.random(shape)
    elif leaf == "weight" and len(shape) == 1 and shape[0] > 1:   # BN / LN scales
        v = 1.0 + 0.2 * r.standard_normal(shape)
    elif shape == (1,):                      # PReLU slopes
        v = np.full(shape, 0.25) + 0.05 * r.standard_normal(shape)
    elif leaf.startswith("bias") or leaf == "in_proj_bias":
        v = 0.05 * r.standard_normal(shape)
    elif name.endswith("stn_fc2.weight"):
        v = 0.02 * r.standard_normal(shape)
    elif name.endswith("stn_fc2.bias"):
        assert ref is not None
        return ref.clone() + torch.from_numpy((0.01 * r.standard_normal(shape)).astype(np.float32))
    elif name.endswith("init_factor.weight"):
        v = r.standard_normal(shape)
    else:                                    # conv / linear / GRU weights
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        a = 1.0 / np.sqrt(fan_in)
        v = r.uniform(-a, a, shape) * 1.7
    return torch.from_numpy(np.asarray(v, dtype=np.float32))


def fill_state(state: Mapping[str, torch.Tensor], seed: int = 0) -> Dict[str, torch.Tensor]:
    """New state dict with the same keys/shapes, procedurally valued."""
    return {k: tensor_for(k, v.shape, seed, ref=v) for k, v in state.items()}


def make_inputs(B: int, K: int = 37, seed: int = 0, H: int = 16, W: int = 64):
    """Synthetic TextZoom-shaped batch: hr [B,4,2H,2W], lr [B,4,H,W], tp [B,K,1,26]."""
    r = _rng("inputs", seed * 1000 + B)
    yy, xx = np.meshgrid(np.arange(2 * H), np.arange(2 * W), indexing="ij")
    hr = np.zeros((B, 4, 2 * H, 2 * W), np.float32)
    for b in range(B):
        for c in range(3):
            f1, f2, ph = r.uniform(0.05, 0.6), r.uniform(0.05, 0.9), r.uniform(0, 6.28)
            hr[b, c] = 0.5 + 0.35 * np.sin(f1 * xx + ph) * np.cos(f2 * yy) + 0.1 * r.standard_normal((2 * H, 2 * W))
    hr[:, :3] = np.clip(hr[:, :3], 0, 1)

    def with_mask(img):
        gray = 0.299 * img[:, 0] + 0.587 * img[:, 1] + 0.114 * img[:, 2]      # dataset/dataset.py:318-323
        img[:, 3] = (gray <= gray.mean((1, 2), keepdims=True)).astype(np.float32)
        return img

    hr = with_mask(hr)
    lr = np.zeros((B, 4, H, W), np.float32)
    lr[:, :3] = hr[:, :3].reshape(B, 3, H, 2, W, 2).mean((3, 5))              # box downsample (synthetic)
    lr = with_mask(lr)
    logits = 3.0 * r.standard_normal((B, 26, K)).astype(np.float32)
    tp = torch.softmax(torch.from_numpy(logits), -1).permute(0, 2, 1).unsqueeze(2).contiguous()  # [B,K,1,26]
    return torch.from_numpy(hr), torch.from_numpy(lr), tp


def planted_logits(seed: int, T: int = 26, B: int = 16, K: int = 37):
    """Recogniser logits [T,B,K] whose per-step argmax follows a planted class sequence with runs and blanks, plus exact ties at
    every fourth step (the first index must win).  Returns (logits float32 ndarray, planted indices [T,B])."""
    rng = np.random.RandomState(seed)
    idx = np.zeros((T, B), np.int64)
    for b in range(B):
        t = 0
        while t < T:
            c = 0 if rng.rand() < 0.35 else rng.randint(1, K)
            run = rng.randint(1, 4)
            idx[t:t + run, b] = c
            t += run
    x = rng.randn(T, B, K).astype(np.float32)
    x[np.arange(T)[:, None], np.arange(B)[None, :], idx] = 9.0
    tie = (idx + 5) % K
    tt, bb = np.nonzero((np.arange(T)[:, None] % 4 == 0) & np.ones((1, B), bool))
    x[tt, bb, tie[tt, bb]] = 9.0
    return x, idx


def random_images(seed: int, n: int):
    """n uint8 RGB arrays [h, w, 3] of TextZoom-like varied sizes (both larger and smaller than 32x128 / 16x64, plus the exact
    target sizes so that the skipped-pass cases occur): smooth strokes + noise, procedurally generated."""
    rng = np.random.RandomState(seed)
    sizes = [(32, 128), (16, 64), (32, 90), (21, 128), (3, 5), (1, 1)]
    while len(sizes) < n:
        sizes.append((int(rng.randint(6, 120)), int(rng.randint(12, 420))))
    out = []
    for h, w in sizes[:n]:
        yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        img = np.zeros((h, w, 3))
        for c in range(3):
            img[..., c] = 128 + 100 * np.sin(rng.uniform(0.05, 0.8) * xx + rng.uniform(0, 6)) * np.cos(rng.uniform(0.05, 0.9) * yy)
        img += rng.randn(h, w, 3) * rng.choice([2.0, 25.0, 80.0])
        out.append(np.clip(img, 0, 255).astype(np.uint8))
    return out
