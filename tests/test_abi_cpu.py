"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/tatt_hip.h declares (no compute calls: there is no GPU here), argument validation reports errors
through tatt_last_error, and the module shell exposes the reference's state_dict contract."""
import ctypes
import os

import pytest
import torch

from tatt_amd import lib as tl


def test_library_exports_every_declared_symbol():
    decls = tl.parse_header()
    assert len(decls) >= 45
    cdll = ctypes.CDLL(tl.LIB_PATH)
    for name in decls:
        assert hasattr(cdll, name), f"{name} declared in include/tatt_hip.h but not exported"
    assert tl.lib().version() >= 100


def test_argument_errors_are_reported_not_thrown():
    L = tl.lib()
    with pytest.raises(RuntimeError, match="tatt_gemm"):
        L.gemm(None, None, None, None, 0, 0, 0, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 1, None, None)
    with pytest.raises(RuntimeError, match="26 keys"):
        L.attn_fwd(None, None, None, None, None, 1, 8, 25, 0, 0, 1.0, None, None)
    assert L.gemm_workspace(64, 64, 4) == 64 * 64 * 4


def test_module_state_dict_contract():
    from tatt_amd.model.tsrn import TSRN_TL_TRANS
    m = TSRN_TL_TRANS(scale_factor=2, width=128, height=32, STN=True, mask=True, srb_nums=5, hidden_units=32, text_embedding=37)
    sd = m.state_dict()
    assert len(sd) == 304 and len(list(m.parameters())) == 245
    assert sum(p.numel() for p in m.parameters()) == 7608334
    shapes = {"block1.0.weight": (64, 4, 9, 9), "block3.gru1.conv1.weight": (64, 128, 1, 1), "block3.gru2.gru.weight_hh_l0_reverse": (96, 32),
              "infoGen.fc_in.weight": (64, 37), "infoGen.transformer.gru_encoding.weight_ih_l0": (1536, 1024),
              "infoGen.transformer.decoder.layers.1.multihead_attn.in_proj_weight": (192, 64), "infoGen.pe.pe": (1, 5000, 64),
              "infoGen.init_factor.weight": (1024, 64), "block8.0.conv.weight": (256, 64, 3, 3), "block8.1.weight": (4, 64, 9, 9),
              "tps.inverse_kernel": (23, 23), "tps.target_coordinate_repr": (1024, 23), "stn_head.stn_fc2.weight": (40, 512),
              "stn_head.stn_convnet.10.0.weight": (256, 256, 3, 3), "block7.1.running_var": (64,)}
    for k, s in shapes.items():
        assert tuple(sd[k].shape) == s, k
    assert sum(p.numel() for p in TSRN_TL_TRANS(text_embedding=2448).parameters()) == 7762638
    # STN init: zero fc2 weight, identity control points (model/stn_head.py:73-90)
    assert float(sd["stn_head.stn_fc2.weight"].abs().max()) == 0.0
    assert abs(float(sd["stn_head.stn_fc2.bias"][0]) - 0.01) < 1e-7


def test_cpu_forward_fails_loudly():
    from tatt_amd.model.tsrn import TSRN_TL_TRANS
    m = TSRN_TL_TRANS()
    with pytest.raises(RuntimeError, match="GPU only"):
        m(torch.rand(1, 4, 16, 64))
