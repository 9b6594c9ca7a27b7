"""Input transform (resizeNormalize / alignCollate, dataset/dataset.py:282-325, :677-693).
CPU: the numpy oracle (oracle/pil_ref.py) against the committed Pillow-made fixture and, where Pillow is importable, against
Pillow itself on fresh sizes.  GPU: the kernel against fixture and oracle, bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import pil_ref as P
from oracle import procgen

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden():
    g = np.load(os.path.join(GOLD, "resize_normalize.npz"))
    return {k: g[k + "8"].astype(np.float32) / np.float32(255) for k in ("hr", "lr")}


def test_oracle_vs_pillow_fixture():
    g, imgs = golden(), procgen.random_images(7, 12)
    hr, lr = P.align_collate(imgs, imgs, 32, 128, 2, True)
    assert np.array_equal(hr, g["hr"]) and np.array_equal(lr, g["lr"])


def test_oracle_vs_pillow_live():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.RandomState(3)
    for it in range(25):
        h, w = int(rng.randint(1, 100)), int(rng.randint(1, 500))
        a = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        for size in ((128, 32), (64, 16), (100, 32)):
            assert np.array_equal(np.asarray(Image.fromarray(a).resize(size, Image.BICUBIC)), P.resize_bicubic(a, size)), (h, w, size)


@pytest.mark.gpu
def test_kernel_vs_fixture_and_oracle():
    from tatt_amd.dataset import alignCollate, resizeNormalize
    g, imgs = golden(), procgen.random_images(7, 12)
    col = alignCollate(imgH=32, imgW=128, down_sample_scale=2, mask=True)
    hr, lr, labels = col([(a, a, "w%d" % i) for i, a in enumerate(imgs)])
    assert labels == tuple("w%d" % i for i in range(12))
    assert np.array_equal(hr.cpu().numpy(), g["hr"]) and np.array_equal(lr.cpu().numpy(), g["lr"])
    # fresh sizes, no mask, one image at a time and as a ragged batch; a tall/wide image near the LDS limit
    more = procgen.random_images(21, 20) + [np.random.RandomState(1).randint(0, 256, (250, 900, 3)).astype(np.uint8)]
    t = resizeNormalize((100, 32), mask=False)
    got = t(more).cpu().numpy()
    for i, a in enumerate(more):
        want = P.resize_normalize(a, (100, 32), mask=False)
        assert np.array_equal(got[i], want), i
        if i % 5 == 0:
            assert np.array_equal(t(a).cpu().numpy(), want)
    # PIL images go in unchanged
    Image = pytest.importorskip("PIL.Image")
    one = resizeNormalize((64, 16), mask=True)(Image.fromarray(imgs[6]))
    assert np.array_equal(one.cpu().numpy(), P.resize_normalize(imgs[6], (64, 16), True))


@pytest.mark.gpu
def test_kernel_refuses_what_does_not_fit():
    from tatt_amd.dataset import resizeNormalize
    big = np.zeros((600, 300, 3), np.uint8)                   # 600 rows x 128 x 3 of intermediate image > 160 KB of LDS
    with pytest.raises(RuntimeError, match="too large"):
        resizeNormalize((128, 32), mask=True)(big)
