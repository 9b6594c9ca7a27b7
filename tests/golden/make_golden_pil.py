"""Generate tests/golden/resize_normalize.npz with Pillow (the library the reference's input transform calls), build container.
Usage:  python tests/golden/make_golden_pil.py

dataset/dataset.py cannot be imported here (lmdb / imgaug / cv2 at module scope); the ten lines of
resizeNormalize.__call__ (:282-325) are restated below on the same PIL calls, with torchvision's ToTensor written out
(uint8 -> float32 / 255).  Inputs are procedural (oracle/procgen.random_images); the fixture holds expected outputs only."""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import procgen                              # noqa: E402


def resize_normalize(arr, size, mask=True):
    img = Image.fromarray(arr).resize(size, Image.BICUBIC)
    t = np.asarray(img).transpose(2, 0, 1).astype(np.float32) / np.float32(255)
    if mask:
        m = img.convert("L")
        thres = np.array(m).mean()
        m = m.point(lambda x: 0 if x > thres else 255)
        t = np.concatenate([t, np.asarray(m)[None].astype(np.float32) / np.float32(255)], 0)
    return t


if __name__ == "__main__":
    imgs = procgen.random_images(7, 12)
    out = {"hr": np.stack([resize_normalize(a, (128, 32)) for a in imgs]), "lr": np.stack([resize_normalize(a, (64, 16)) for a in imgs])}
    out["hr8"] = np.round(out["hr"] * 255).astype(np.uint8)
    out["lr8"] = np.round(out["lr"] * 255).astype(np.uint8)
    for k in ("hr", "lr"):                              # stored as bytes: x / 255 in float32 is recomputed by the test
        assert np.array_equal(out[k + "8"].astype(np.float32) / np.float32(255), out[k])
        del out[k]
    fn = os.path.join(HERE, "resize_normalize.npz")
    np.savez_compressed(fn, **out)
    print(fn, os.path.getsize(fn) // 1024, "KiB", {k: v.shape for k, v in out.items()}, [a.shape for a in imgs])
