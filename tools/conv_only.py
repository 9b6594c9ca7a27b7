"""Runs only the SRB-shaped packed conv (gemm_kernel<OP_IM2COL,OP_CONVW_P,FAST>, 3x3 64->64 on [64,16,64]) a few times:
target for the rocprofv3 --pmc passes that measure its HBM-side traffic (profiles/r01_conv_traffic.md)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tatt_amd.lib import lib
L = lib(); B = 64; R = B * 1024
x, y, w, b = [torch.randn(*s, device="cuda") for s in ((R, 64), (R, 64), (64, 64, 3, 3), (64,))]
wf, wd = torch.empty(64 * 576, device="cuda"), torch.empty(64 * 576, device="cuda")
L.conv_pack_w(w, wf, wd, 64, 64, 3, 3)
fl = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # 0 fp32 MFMA, 2 TATT_BF16, 4 TATT_F32X3, 10 = TATT_BF16 | TATT_X_BF16 (input stored as bf16)
if fl & 8:
    x = x.bfloat16()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    L.conv_fwd(x, wf, b, y, B, 16, 64, 64, 64, 3, 3, 64, 64, 576, 1, 0, 64, 0, fl)
torch.cuda.synchronize()
