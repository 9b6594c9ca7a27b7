"""CPU oracle of the reference's input transform.  TEST INFRASTRUCTURE ONLY: imported by tests/, never
by the product package.  Own restatement of

  * resizeNormalize.__call__   dataset/dataset.py:282-325   (PIL resize BICUBIC -> ToTensor -> gray-mean mask channel)
  * alignCollate.__call__      dataset/dataset.py:677-693   (HR to (imgW, imgH), LR to (imgW/scale, imgH/scale), stacked)

The arithmetic of `img.resize(size, Image.BICUBIC)` and `img.convert('L')` lives in Pillow, a third-party dependency that is
not vendored under /root/reference and that the reference does not pin (README: "pip install pillow"); this container has
Pillow 12.2.0.  Its published algorithm (libImaging/Resample.c: separable two-pass resampling, horizontal then vertical,
support-scaled bicubic a = -0.5, coefficients normalised in double and quantised to 22 fractional bits, 8-bit intermediate;
libImaging/Convert.c: L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16) is restated below in numpy integers and pinned
bit-exactly against Pillow itself (tests/test_input_pipeline.py, and tests/golden/resize_normalize.npz made with Pillow by
tests/golden/make_golden_pil.py)."""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the full-image box: (xmin [out], xcount [out], kk [out, ksize] int)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int64)
    kk = np.zeros((out_size, ksize), np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img: np.ndarray, out_size: int) -> np.ndarray:
    """One resampling pass along axis 1 of a [rows, n, C] uint8 array."""
    bounds, kk = precompute_coeffs(img.shape[1], out_size)
    out = np.zeros((img.shape[0], out_size, img.shape[2]), np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        xmin, xmax = bounds[xx]
        acc = (src[:, xmin:xmin + xmax, :] * kk[xx, :xmax, None]).sum(1) + (1 << (PRECISION_BITS - 1))
        out[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_bicubic(img: np.ndarray, size) -> np.ndarray:
    """PIL.Image.resize(size=(W, H), Image.BICUBIC) of a uint8 [h, w, C] array; a pass over an unchanged dimension is skipped."""
    W, H = size
    if img.shape[1] != W:
        img = _pass(img, W)
    if img.shape[0] != H:
        img = _pass(img.transpose(1, 0, 2), H).transpose(1, 0, 2)
    return img


def resize_normalize(img: np.ndarray, size, mask: bool = True) -> np.ndarray:
    """resizeNormalize.__call__ (dataset/dataset.py:282-325, ratio_keep=False, no aug): uint8 RGB [h, w, 3] -> float32 [3 or 4, H, W]."""
    r = resize_bicubic(img, size)
    out = [r.transpose(2, 0, 1).astype(np.float32) / np.float32(255)]
    if mask:
        v = r.astype(np.int64)
        L = (v[..., 0] * 19595 + v[..., 1] * 38470 + v[..., 2] * 7471 + 0x8000) >> 16
        thres = L.astype(np.uint8).mean()
        out.append(np.where(L > thres, 0, 255)[None].astype(np.float32) / np.float32(255))
    return np.concatenate(out, 0)


def align_collate(images_hr, images_lr, imgH=32, imgW=128, down_sample_scale=2, mask=True):
    """alignCollate.__call__ (dataset/dataset.py:677-693): lists of uint8 RGB arrays -> (hr [B,4,imgH,imgW], lr [B,4,imgH/s,imgW/s])."""
    hr = np.stack([resize_normalize(i, (imgW, imgH), mask) for i in images_hr])
    lr = np.stack([resize_normalize(i, (imgW // down_sample_scale, imgH // down_sample_scale), mask) for i in images_lr])
    return hr, lr
