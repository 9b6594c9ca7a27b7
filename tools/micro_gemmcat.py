"""The GruBlock's folded input projection and its input-gradient GEMMs at B=64, stand-alone (bash tools/run_micro.sh tools/micro_gemmcat.py):
fwd  gi[R,192] = [x1|x2][R,64|128] Wf^T + bf            (tatt_gemm_cat, two A sources)
bwd  [dx1|dx2] = dgi[R,192] Wf[192,Cin]                  (tatt_gemm_cat, two destinations; tatt_gemm_cat_bnbwd with the BatchNorm epilogue)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tatt_amd.lib import lib
L = lib()
R = 65536
g = lambda *s: torch.randn(*s, device="cuda") * 0.3
x1, x2, gi, dgi = g(R, 64), g(R, 64), torch.empty(R, 192, device="cuda"), g(R, 192)
dx1, dx2 = torch.empty(R, 64, device="cuda"), torch.zeros(R, 64, device="cuda")
stat = torch.stack([torch.zeros(64), torch.ones(64)]).cuda()
ga, be = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
part = torch.empty(R // 128 * 2 * 64, device="cuda")
for Cin in (64, 128):
    Wf, bf = g(192, Cin), g(192)
    two = Cin == 128
    for _ in range(6):
        L.gemm_cat(x1, x2 if two else None, 64, Wf, gi, None, 0, bf, R, 192, Cin, 64, Cin, 1, 192, 0, 0, 0, 0)
        L.gemm_cat(dgi, None, 0, Wf, dx1, dx2 if two else None, 64, None, R, Cin, 192, 192, 1, Cin, 64, 64, 0, 1 if two else 0, 0)
        L.gemm_cat_bnbwd(dgi, None, 0, Wf, dx1, dx2 if two else None, 64, None, R, Cin, 192, 192, 1, Cin, 64, 64, 0, 1 if two else 0, 0,
                         x1, 64, stat, ga, be, 0, 64, part)
    torch.cuda.synchronize()
# the decoder's 64 -> 64 linears (forward and input gradient)
W = g(64, 64)
for _ in range(6):
    L.gemm(x1, W, dx1, None, R, 64, 64, 64, 1, 64, 1, 64, 1, 0, 0, 0, 0, 0, 1, None)
    L.gemm(x1, W, dx1, None, R, 64, 64, 64, 1, 1, 64, 64, 1, 0, 0, 0, 0, 0, 1, None)
torch.cuda.synchronize()
