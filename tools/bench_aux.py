"""Timings of the §8(f) kernels around the hot path: input transform, CRNN prior, CTC decode, eval metrics (B=64).
python tools/bench_aux.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tatt_amd.dataset import pack_images
from tatt_amd.lib import lib
from tatt_amd.model.crnn import CRNN
from tatt_amd.utils.metrics import ctc_greedy
from tatt_amd.utils.ssim_psnr import SSIM, calculate_psnr


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


if __name__ == "__main__":
    B = 64
    rng = np.random.RandomState(0)
    imgs = [rng.randint(0, 256, (int(rng.randint(24, 64)), int(rng.randint(80, 260)), 3)).astype(np.uint8) for _ in range(B)]
    buf, offs, hs, ws = pack_images(imgs)
    d = [t.cuda() for t in (buf, offs, hs, ws)]
    out = torch.empty(B, 4, 32, 128, device="cuda")
    L = lib()
    mh, mw = int(hs.max()), int(ws.max())
    t = timeit(lambda: L.resize_normalize(*d, out, B, 32, 128, 1, mh, mw))
    nbytes = buf.numel() + out.numel() * 4
    print(f"resize_normalize HR  B=64: {t:7.1f} us  ({nbytes / t / 1e3:.1f} GB/s of {nbytes / 1e6:.2f} MB in+out; one workgroup per image)")
    m = CRNN(32, 1, 37, 256).cuda().eval()
    hr = torch.rand(B, 4, 32, 128, device="cuda")
    with torch.no_grad():
        t = timeit(lambda: m.forward_from_images(hr), 20)
        logits = m.forward_from_images(hr)
    print(f"CRNN prior forward   B=64: {t:7.1f} us")
    t = timeit(lambda: ctc_greedy(logits))
    print(f"ctc_greedy           B=64: {t:7.1f} us")
    sr = torch.rand(B, 4, 32, 128, device="cuda")
    ss = SSIM()
    t = timeit(lambda: (ss(sr, hr), calculate_psnr(sr, hr)))
    print(f"SSIM + PSNR          B=64: {t:7.1f} us")
