"""Kernel timing of the GruBlock-shaped weight-gradient group (4 and 6 problems over 65,536 rows); run under rocprofv3 --kernel-trace
(host-side timing is launch-bound): python tools/micro_tn.py"""
import ctypes
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tatt_amd.lib import lib
L = lib()
R = 65536
dgi, dghn, hp = (torch.randn(R, 192, device="cuda"), torch.randn(R, 64, device="cuda"), torch.randn(R, 64, device="cuda"))
x1, x2 = torch.randn(R, 64, device="cuda"), torch.randn(R, 64, device="cuda")
ws = torch.empty(L.tn_small_workspace(96, 64), device="cuda")


def group(nin):
    Cin = 64 * nin
    dWf, dbf = torch.zeros(192 * Cin, device="cuda"), torch.zeros(192, device="cuda")
    dWhh, dbhh = torch.zeros(2, 96 * 32, device="cuda"), torch.zeros(2, 96, device="cuda")
    pr = []
    for d in range(2):
        pr.append((dgi.view(-1)[96 * d:], hp.view(-1)[32 * d:], dWhh[d], dbhh[d], R, 96, 32, 192, 64, 32, dghn.view(-1)[32 * d:]))
        for s, x in enumerate([x1, x2][:nin]):
            pr.append((dgi.view(-1)[96 * d:], x, dWf[96 * d * Cin + 64 * s:], dbf[96 * d:] if s == 0 else None, R, 96, 64, 192, 64, Cin))
    n = len(pr)
    VP, LG, IN = ctypes.c_void_p * n, ctypes.c_long * n, ctypes.c_int * n
    args = (n, VP(*[p[0].data_ptr() for p in pr]), LG(*[p[7] for p in pr]), VP(*[p[1].data_ptr() for p in pr]), LG(*[p[8] for p in pr]),
            VP(*[p[2].data_ptr() for p in pr]), LG(*[p[9] for p in pr]), VP(*[None if p[3] is None else p[3].data_ptr() for p in pr]),
            IN(*[p[5] for p in pr]), IN(*[p[6] for p in pr]), IN(*[p[4] for p in pr]),
            VP(*[p[10].data_ptr() if len(p) > 10 else None for p in pr]), LG(*[64] * n), IN(*[64] * n))
    for _ in range(10):
        L.tn_small_group(*args, ws)
    torch.cuda.synchronize()
    # check against torch
    ref = (dgi[:, :96].double().T @ x1.double()) * 10
    got = dWf.view(192, Cin)[:96, :64].double()
    print(nin, "rel err", float((got - ref).norm() / ref.norm()))


group(1)
group(2)
