"""CRNN text-prior generator (forward) on the GPU against the reference goldens and the oracle."""
import copy
import os

import numpy as np
import pytest
import torch

from oracle import crnn_ref as C
from oracle import procgen
from tests.util import rel_l2

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def build(train):
    from tatt_amd.model.crnn import CRNN
    m = CRNN(imgH=32, nc=1, nclass=37, nh=256)
    m.load_state_dict(procgen.fill_state(m.state_dict(), seed=3), strict=True)
    return m.cuda().train(train)


@pytest.mark.parametrize("name,train", [("crnn_B2_eval", False), ("crnn_B3_train", True)])
def test_crnn_forward_vs_reference_golden(name, train):
    from tatt_amd.model.crnn import parse_crnn_data
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    B = gold["hr/gray"].shape[0]
    hr, lr, _ = procgen.make_inputs(B, 37, seed=4)
    m = build(train)                                         # one module, hr then lr: as the golden generator ran it
    for tag, img in (("hr", hr), ("lr", lr)):
        if not train:
            m = build(train)
        gray = parse_crnn_data(img.cuda())
        assert float(np.abs(gray.cpu().numpy() - gold[tag + "/gray"]).max()) < 2e-6
        with torch.no_grad():
            logits = m(gray)
        assert logits.shape == (26, B, 37)
        assert rel_l2(logits.cpu().numpy(), gold[tag + "/logits"]) < 2e-5, tag
    if train:
        sd = m.state_dict()
        for k in ("cnn.batchnorm2.running_mean", "cnn.batchnorm6.running_var"):
            assert rel_l2(sd[k].cpu().numpy(), gold["bn/" + k]) < 1e-5
        assert int(sd["cnn.batchnorm4.num_batches_tracked"]) == 2


def test_crnn_from_images_b64_matches_oracle_and_feeds_the_sr_net():
    """B=64 (the benchmark batch): fused prep+forward vs the fp32 oracle; the prior plugs into TSRN_TL_TRANS."""
    from tatt_amd.model.crnn import text_prior
    from tatt_amd.model.tsrn import TSRN_TL_TRANS
    B = 64
    hr, lr, _ = procgen.make_inputs(B, 37, seed=6)
    m = build(False)
    with torch.no_grad():
        logits = m.forward_from_images(lr.cuda())
    P = procgen.fill_state(m.state_dict(), seed=3)
    ref = C.crnn_forward(P, C.parse_crnn_data(lr), training=False)
    assert rel_l2(logits.cpu().numpy(), ref.numpy()) < 2e-5
    prior = text_prior(logits)
    assert prior.shape == (B, 37, 1, 26) and not prior.is_contiguous()
    sr = TSRN_TL_TRANS(text_embedding=37).cuda().eval()(lr.cuda(), prior.detach())
    assert sr.shape == (B, 4, 32, 128) and torch.isfinite(sr).all()
