"""Evaluation path on the GPU: greedy CTC decode, str_filt accuracy, bicubic LR upsample and the
per-batch PSNR / SSIM bookkeeping of TextSR.eval, against the reference-made goldens and the CPU oracle."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import crnn_ref as C
from oracle import procgen
from tests.util import make_state

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_get_string_crnn_vs_reference_golden():
    """Strings made by the reference's get_string_crnn on planted logits (runs, blanks, exact ties): bit-exact."""
    from tatt_amd.utils.metrics import ctc_greedy, get_string_crnn
    gold = json.load(open(os.path.join(GOLD, "ctc_strings.json")))
    for case in gold["cases"]:
        x, idx = procgen.planted_logits(case["seed"])
        xg = torch.from_numpy(x).cuda()
        assert get_string_crnn(xg, False) == case["strings"]
        arg, codes, lens = ctc_greedy(xg)
        assert torch.equal(arg.cpu().long(), torch.from_numpy(x).argmax(-1).t())
        # ties resolve to the first index: the planted class unless the tie partner (idx + 5) % K has wrapped below it
        T = idx.shape[0]
        tie = (np.arange(T)[:, None] % 4 == 0)
        want = np.where(tie, np.minimum(idx, (idx + 5) % 37), idx)
        assert np.array_equal(arg.cpu().numpy().T, want)
        for b, s in enumerate(case["strings"]):
            assert "".join(C.ALPHABET[i] for i in codes[b, :lens[b]].tolist()) == s
            assert int(codes[b, lens[b]:].abs().sum()) == 0


@pytest.mark.parametrize("T,B,K", [(26, 1, 37), (7, 130, 2448), (33, 5, 64)])
def test_get_string_crnn_shapes_vs_oracle(T, B, K):
    """Ragged class counts (K < / = / > the wave width), a single sample, all-blank and all-equal rows."""
    from tatt_amd.utils.metrics import ctc_greedy, get_string_crnn
    x, _ = procgen.planted_logits(5, T, B, K)
    x[:, 0, :] = 0.0                                          # all-equal logits: argmax 0 everywhere -> empty string
    alphabet = "-" + "".join(chr(0x4E00 + i) for i in range(K - 1))
    got = get_string_crnn(torch.from_numpy(x).cuda(), False, alphabet)
    assert got == C.get_string_crnn(torch.from_numpy(x), alphabet)
    assert got[0] == ""
    dup = ("-" + "ab" * K)[:K]                                # repeated characters: the reference merges by character
    ref = []
    for row in torch.from_numpy(x).permute(1, 0, 2).argmax(-1).tolist():
        s, last = "", ""
        for i in row:
            if dup[i] != last:
                if i != 0:
                    s += dup[i]
                    last = dup[i]
                else:
                    last = ""
        ref.append(s)
    assert get_string_crnn(torch.from_numpy(x).cuda(), False, dup) == ref


@pytest.mark.parametrize("shape,size", [((5, 4, 16, 64), (32, 128)), ((2, 3, 32, 128), (32, 100)), ((1, 1, 7, 9), (5, 30))])
def test_bicubic_resize_vs_torch(shape, size):
    from tatt_amd.evaluate import bicubic_resize
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(1))
    want = F.interpolate(x, size, mode="bicubic")
    got = bicubic_resize(x.cuda(), size).cpu()
    assert (got - want).abs().max().item() < 2e-6


def test_evaluate_matches_oracle():
    """Two ragged batches through tatt_amd.evaluate.evaluate vs oracle/crnn_ref.eval_batches: identical strings and accuracies,
    PSNR within 1e-3 dB, SSIM within 1e-5 (SURVEY §8c tolerances)."""
    from tatt_amd.evaluate import evaluate
    from tatt_amd.model.crnn import CRNN
    from tatt_amd.model.tsrn import TSRN_TL_TRANS
    P_sr = make_state(37)
    sr_net = TSRN_TL_TRANS(scale_factor=2, width=128, height=32, STN=True, srb_nums=5, mask=True, hidden_units=32)
    sr_net.load_state_dict(P_sr, strict=True)
    mods, Ps = [], []
    for seed in (3, 8):                                        # student prior generator, test recogniser
        m = CRNN(imgH=32, nc=1, nclass=37, nh=256)
        sd = procgen.fill_state(m.state_dict(), seed=seed)
        m.load_state_dict(sd, strict=True)
        mods.append(m.cuda())
        Ps.append(sd)
    batches = []
    for B, seed in ((5, 0), (3, 1)):
        hr, lr, _ = procgen.make_inputs(B, 37, seed=seed)
        batches.append([hr, lr, None])
    # labels: the oracle's own HR predictions for every other sample (so that the accuracies are not trivially 0), else junk
    with torch.no_grad():
        for bt in batches:
            s = C.get_string_crnn(C.crnn_forward(Ps[1], C.parse_crnn_data(bt[0]), training=False))
            bt[2] = [w.upper() + "!" if i % 2 == 0 else "zz9" for i, w in enumerate(s)]
    want = C.eval_batches(P_sr, Ps[0], Ps[1], batches)
    got = evaluate(sr_net.cuda(), mods[0], mods[1], batches)
    for k in ("strings_lr", "strings_hr", "strings_sr"):
        assert got[k] == want[k], k
    for k in ("accuracy", "accuracy_lr", "accuracy_hr"):
        assert got[k] == want[k], k
    assert want["accuracy_hr"] >= 0.5
    for k, tol in (("psnr", 1e-3), ("psnr_lr", 1e-3), ("ssim", 1e-5), ("ssim_lr", 1e-5)):
        for a, b in zip(got[k], want[k]):
            assert abs(float(a) - float(b)) < tol, (k, float(a), float(b))
    assert abs(got["psnr_avg"] - want["psnr_avg"]) < 1e-3 and abs(got["ssim_avg"] - want["ssim_avg"]) < 1e-5
