"""Checkpoint format: files written by tatt_amd.checkpoint load into the REAL reference modules and
files laid out as the reference's save_checkpoint writes them load here — the former only where /root/reference exists."""
import os
import sys
import types

import pytest
import torch

from oracle import procgen
from tatt_amd import checkpoint as ck
from tatt_amd.model.crnn import CRNN
from tatt_amd.model.tsrn import TSRN_TL_TRANS

HAVE_REF = os.path.isdir("/root/reference/model")


def nets():
    m = TSRN_TL_TRANS(scale_factor=2, width=128, height=32, STN=True, mask=True, srb_nums=5, hidden_units=32, text_embedding=37)
    m.load_state_dict(procgen.fill_state(m.state_dict(), seed=5), strict=True)
    c = CRNN(32, 1, 37, 256)
    c.load_state_dict(procgen.fill_state(c.state_dict(), seed=6), strict=True)
    return m, c


def test_round_trip_and_layout(tmp_path):
    m, c = nets()
    d = str(tmp_path / "ckpt")
    ck.save_checkpoint([m], 3, 1200, {"easy": 0.5}, {"easy": {"accuracy": 0.5, "psnr": 21.0}}, True, [0.1, 0.2], recognizer=[c],
                       ckpt_path=d, batch_size=64)
    assert sorted(os.listdir(d)) == ["model_1200.pth", "model_best.pth", "recognizer_1200.pth", "recognizer_best.pth"]
    raw = torch.load(os.path.join(d, "model_1200.pth"), weights_only=True)
    assert set(raw) == {"state_dict_G", "info", "best_history_res", "best_model_info", "param_num", "converge"}
    assert raw["info"] == {"arch": "tatt", "iters": 1200, "epochs": 3, "batch_size": 64, "voc_type": "eng", "up_scale_factor": 2}
    assert raw["param_num"] == sum(p.numel() for p in m.parameters()) == 7608334
    for resume in (d, os.path.join(d, "model_1200.pth")):                  # directory -> model_best.pth, or a file
        got = ck.generator_init(resume, device="cpu")
        assert (got["epoch"], got["iter"]) == (3, 1200)
        for (k, a), (k2, b) in zip(m.state_dict().items(), got["model"].state_dict().items()):
            assert k == k2 and torch.equal(a, b)
    c2 = ck.CRNN_init(os.path.join(d, "recognizer_best.pth"), device="cpu")
    assert all(torch.equal(a, b) for a, b in zip(c.state_dict().values(), c2.state_dict().values()))
    assert ck.generator_init("", voc_type="all", device="cpu")["model"].infoGen.fc_in.in_features == 2448


@pytest.mark.skipif(not HAVE_REF, reason="needs the reference checkout (build container only)")
def test_files_interchange_with_reference_modules(tmp_path):
    for n in ["IPython", "cv2", "torchvision", "torchvision.models", "torchvision.transforms"]:
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules["IPython"].embed = lambda *a, **k: None
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.path.insert(0, "/root/reference")
    try:
        from model import tsrn as ref_tsrn
        from model.crnn import crnn as ref_crnn
    finally:
        sys.path.remove("/root/reference")
    m, c = nets()
    d = str(tmp_path / "ckpt")
    ck.save_checkpoint([m], 1, 10, {}, {}, True, [], recognizer=[c], ckpt_path=d, batch_size=64)
    # ours -> reference (what generator_init :203-213 and CRNN_init :334-343 do)
    ref = ref_tsrn.TSRN_TL_TRANS(scale_factor=2, width=128, height=32, STN=True, mask=True, srb_nums=5, hidden_units=32,
                                 text_embedding=37)
    ref.load_state_dict(torch.load(os.path.join(d, "model_best.pth"), weights_only=True)["state_dict_G"])
    rc = ref_crnn.CRNN(imgH=32, nc=1, nclass=37, nh=256)
    rc.load_state_dict(torch.load(os.path.join(d, "recognizer_best.pth"), weights_only=True))
    # reference -> ours: a dict assembled as save_checkpoint :252-260 assembles it, from the reference modules' state_dicts
    ref.load_state_dict(procgen.fill_state(ref.state_dict(), seed=9), strict=True)
    save_dict = {"state_dict_G": ref.state_dict(),
                 "info": {"arch": "tatt", "iters": 7, "epochs": 2, "batch_size": 64, "voc_type": "eng", "up_scale_factor": 2},
                 "best_history_res": {"easy": 0.1}, "best_model_info": {}, "param_num": sum(p.nelement() for p in ref.parameters()),
                 "converge": []}
    f = str(tmp_path / "model_7.pth")
    torch.save(save_dict, f)
    torch.save(rc.state_dict(), str(tmp_path / "recognizer_7.pth"))
    got = ck.generator_init(f, device="cpu")
    assert got["iter"] == 7 and all(torch.equal(a, b) for a, b in zip(ref.state_dict().values(), got["model"].state_dict().values()))
    c3 = ck.CRNN_init(str(tmp_path / "recognizer_7.pth"), device="cpu")
    assert list(c3.state_dict()) == list(rc.state_dict())
