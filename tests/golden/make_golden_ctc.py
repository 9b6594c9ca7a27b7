"""Generate tests/golden/ctc_strings.json from the REAL reference decode helpers, build container only.
Usage:  python tests/golden/make_golden_ctc.py

`get_string_crnn` (utils/metrics.py:128-146) and `str_filt` (utils/util.py:14-42) are imported from /root/reference
with empty stand-in modules for two imports they never use (`editdistance`, `IPython`).  The fixture holds
seeds (inputs come from oracle/procgen.planted_logits), the planted index sequences and the expected strings only.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
for name in ("editdistance", "IPython"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["IPython"].embed = lambda *a, **k: None
sys.path.insert(0, "/root/reference")
os.chdir("/root/reference")                              # str_filt opens its vocabulary files by relative path

from utils.metrics import get_string_crnn               # noqa: E402
from utils.util import str_filt                          # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.procgen import planted_logits               # noqa: E402


if __name__ == "__main__":
    out = {"cases": [], "filt": []}
    for seed in (11, 12, 13):
        x, idx = planted_logits(seed)
        out["cases"].append({"seed": seed, "planted": idx.T.tolist(), "strings": get_string_crnn(torch.from_numpy(x))})
    words = ["Hello, World!", "ABC-123_def", "naïve café 42", "", "----", "a1B2c3D4", "한글 text 7", "x" * 30, "Tel:+82-10"]
    for mode, is_crnn in (("digit", False), ("lower", True), ("upper", False), ("lower", False), ("all", False)):
        out["filt"].append({"mode": mode, "is_crnn": is_crnn, "in": words, "out": [str_filt(w, mode, is_crnn) for w in words]})
    fn = os.path.join(HERE, "ctc_strings.json")
    json.dump(out, open(fn, "w"), ensure_ascii=False, indent=0)
    print(fn, os.path.getsize(fn), "bytes")
    print(out["cases"][0]["strings"][:4], out["filt"][1]["out"])
