"""Generate tests/golden/crnn_*.npz from the REAL reference CRNN (model/crnn/crnn.py), build container only.
Usage:  python tests/golden/make_golden_crnn.py

Inputs and weights are procedural (oracle/procgen.py); the fixtures hold expected outputs only.  `parse_crnn_data`
lives in interfaces/base.py, which cannot be imported here (easydict/ptflops/mmcv missing); its five lines
(F.interpolate(..., (32, 100), mode='bicubic') + the gray mix, base.py:441-459) are restated below on the same torch op.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from model.crnn import crnn as ref_crnn                 # noqa: E402
from oracle import procgen                              # noqa: E402


def parse_crnn_data(imgs):
    x = torch.nn.functional.interpolate(imgs, (32, 100), mode="bicubic")
    return 0.299 * x[:, 0:1] + 0.587 * x[:, 1:2] + 0.114 * x[:, 2:3]


def case(B, train):
    m = ref_crnn.CRNN(imgH=32, nc=1, nclass=37, nh=256)
    m.load_state_dict(procgen.fill_state(m.state_dict(), seed=3), strict=True)
    m.train(train)
    hr, lr, _ = procgen.make_inputs(B, 37, seed=4)
    out = {}
    with torch.no_grad():
        for tag, img in (("hr", hr), ("lr", lr)):
            g = parse_crnn_data(img[:, :3])
            out[f"{tag}/gray"] = g.numpy()
            feat = m.cnn(g)
            out[f"{tag}/cnn"] = feat.numpy()
            seq = feat.squeeze(2).permute(2, 0, 1)
            mid = m.rnn[0](seq)
            out[f"{tag}/rnn0"] = mid.numpy()
            out[f"{tag}/logits"] = m.rnn[1](mid).numpy()
    if train:
        sd = m.state_dict()
        for k in ("cnn.batchnorm2.running_mean", "cnn.batchnorm6.running_var"):
            out["bn/" + k] = sd[k].float().numpy()
    return out


if __name__ == "__main__":
    torch.set_num_threads(8)
    for B, train in [(2, False), (3, True)]:
        o = case(B, train)
        fn = os.path.join(HERE, f"crnn_B{B}_{'train' if train else 'eval'}.npz")
        np.savez_compressed(fn, **{k: v.astype(np.float32) for k, v in o.items()})
        print(fn, os.path.getsize(fn) // 1024, "KiB", {k: v.shape for k, v in o.items()})
