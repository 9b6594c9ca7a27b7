"""Per-shape timing of the hot kernels at the BASELINE batch (B=64) with events on the launch stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tatt_amd.lib import lib

L = lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = B * 1024
ws = torch.empty(48 << 20, device="cuda")


def t(name, fn, flops=0, bytes_=0, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"{name:46s} {us:9.1f} us  {flops / us / 1e6:7.1f} TF  {bytes_ / us / 1e6:7.2f} TB/s")


def rnd(*s):
    return torch.randn(*s, device="cuda")


x64, y64, w33, b64 = rnd(R, 64), rnd(R, 64), rnd(64, 64, 3, 3), rnd(64)
t("conv3x3 64->64 fwd", lambda: L.conv_fwd(x64, w33, b64, y64, B, 16, 64, 64, 64, 3, 3, 64, 64, 576, 9, 0, 1, 0, 0), 2 * R * 64 * 576, R * 512)
wf33, wd33 = rnd(64 * 576), rnd(64 * 576)
L.conv_pack_w(w33, wf33, wd33, 64, 64, 3, 3)
t("conv3x3 64->64 fwd packed", lambda: L.conv_fwd(x64, wf33, b64, y64, B, 16, 64, 64, 64, 3, 3, 64, 64, 576, 1, 0, 64, 0, 0), 2 * R * 64 * 576, R * 512)
t("conv_pack_w 64x64x3x3", lambda: L.conv_pack_w(w33, wf33, wd33, 64, 64, 3, 3), 0, 64 * 576 * 12)
t("conv3x3 64->64 dgrad", lambda: L.conv_fwd(x64, w33, None, y64, B, 16, 64, 64, 64, 3, 3, 64, 64, 9, 576, 8, -1, 0, 0), 2 * R * 64 * 576, R * 512)
dw = rnd(64, 64, 3, 3)
for sk in (8, 16, 32, 64):
    t(f"conv3x3 64->64 wgrad sk={sk}", lambda: L.conv_wgrad(x64, y64, dw, B, 16, 64, 64, 64, 3, 3, 64, 64, sk, ws, 1), 2 * R * 64 * 576, R * 512)
y256, w256 = rnd(R, 256), rnd(256, 64, 3, 3)
t("conv3x3 64->256 fwd", lambda: L.conv_fwd(x64, w256, None, y256, B, 16, 64, 64, 256, 3, 3, 64, 256, 576, 9, 0, 1, 0, 0), 2 * R * 256 * 576, R * 1280)
t("conv3x3 256->64 dgrad", lambda: L.conv_fwd(y256, w256, None, x64, B, 16, 64, 256, 64, 3, 3, 256, 64, 9, 576, 8, -1, 0, 0), 2 * R * 256 * 576, R * 1280)
dw256 = rnd(256, 64, 3, 3)
t("conv3x3 64->256 wgrad sk=16", lambda: L.conv_wgrad(x64, y256, dw256, B, 16, 64, 64, 256, 3, 3, 64, 256, 16, ws, 1), 2 * R * 256 * 576, R * 1280)
x4, w99 = rnd(R, 4), rnd(64, 4, 9, 9)
t("conv9x9 4->64 fwd", lambda: L.conv_fwd(x4, w99, b64, y64, B, 16, 64, 4, 64, 9, 9, 4, 64, 324, 81, 0, 1, 0, 0), 2 * R * 64 * 324, R * 272)
up, pre, w994 = rnd(4 * R, 64), rnd(4 * R, 4), rnd(4, 64, 9, 9)
z = torch.empty(L.conv_skinny_workspace(B, 32, 128, 64, 4, 9, 9), device="cuda")
t("conv9x9 64->4 skinny fwd", lambda: L.conv_skinny_fwd(up, w994, None, pre, B, 32, 128, 64, 4, 9, 9, 64, 4, 5184, 81, 0, 1, 0, z), 2 * 4 * R * 4 * 5184, 4 * R * (256 + 2 * 1296))
t("conv9x9 4->64 dgrad (final, Cin_call=4)", lambda: L.conv_fwd(pre, w994, None, up, B, 32, 128, 4, 64, 9, 9, 4, 64, 81, 5184, 80, -1, 0, 0), 2 * 4 * R * 64 * 324, 4 * R * 272)
dw994 = rnd(4, 64, 9, 9)
t("conv9x9 64->4 wgrad sk=32", lambda: L.conv_wgrad(up, pre, dw994, B, 32, 128, 64, 4, 9, 9, 64, 4, 32, ws, 1), 2 * 4 * R * 4 * 5184, 4 * R * 272)
# GEMMs
W64, W192, o192 = rnd(64, 64), rnd(192, 64), rnd(R, 192)
t("gemm NT [R,64]x[64,64]", lambda: L.gemm(x64, W64, y64, b64, R, 64, 64, 64, 1, 64, 1, 64, 1, 0, 0, 0, 0, 0, 1, None), 2 * R * 64 * 64, R * 512)
t("gemm NT [R,64]x[96,64] ldc192", lambda: L.gemm(x64, W192, o192, None, R, 96, 64, 64, 1, 64, 1, 192, 1, 0, 0, 0, 0, 0, 1, None), 2 * R * 96 * 64, R * 640)
t("gemm NN dx [R,64]x[64,64]", lambda: L.gemm(x64, W64, y64, None, R, 64, 64, 64, 1, 1, 64, 64, 1, 0, 0, 0, 0, 0, 1, None), 2 * R * 64 * 64, R * 512)
t("gemm NT [R,64]x[192,64]", lambda: L.gemm(x64, W192, o192, None, R, 192, 64, 64, 1, 64, 1, 192, 1, 0, 0, 0, 0, 0, 1, None), 2 * R * 192 * 64, R * 1024)
t("gemm NN dx [R,192]x[192,64]", lambda: L.gemm(o192, W192, y64, None, R, 64, 192, 192, 1, 1, 64, 64, 1, 0, 0, 0, 0, 0, 1, None), 2 * R * 192 * 64, R * 1024)
t("gemm NT [R,64]x[64,64] acc", lambda: L.gemm(x64, W64, y64, b64, R, 64, 64, 64, 1, 64, 1, 64, 1, 0, 0, 0, 1, 0, 1, None), 2 * R * 64 * 64, R * 768)
t("gemm NN dx [R,96(ld192)]x[96,64]", lambda: L.gemm(o192, W192, y64, None, R, 64, 96, 192, 1, 1, 64, 64, 1, 0, 0, 0, 0, 0, 1, None), 2 * R * 96 * 64, R * 640)
dW = rnd(64, 64)
for sk in (64, 128, 256, 512):
    t(f"gemm TN dW [64,64] K=R sk={sk}", lambda: L.gemm(x64, y64, dW, None, 64, 64, R, 1, 64, 1, 64, 64, 1, 0, 0, 0, 1, 0, sk, ws), 2 * R * 64 * 64, R * 512)
dW96 = rnd(96, 32)
t("gemm TN dW_hh [96,32] K=R sk=256", lambda: L.gemm(o192, x64, dW96, None, 96, 32, R, 1, 192, 1, 64, 32, 1, 0, 0, 0, 1, 0, 256, ws), 2 * R * 96 * 32, R * 512)
h, Wh, gh = rnd(2, 64, 512), rnd(2, 1536, 512), rnd(2, 64, 1536)
for sk in (1, 2, 4, 8):
    t(f"RPE fwd gemm sk={sk}", lambda: L.gemm(h, Wh, gh, None, 64, 1536, 512, 512, 1, 512, 1, 1536, 2, 64 * 512, 1536 * 512, 64 * 1536, 0, 0, sk, ws), 4 * 64 * 1536 * 512, 6.3e6)
for sk in (1, 4, 8, 16):
    t(f"RPE bwd gemm sk={sk}", lambda: L.gemm(gh, Wh, h, None, 64, 512, 1536, 1536, 1, 1, 512, 512, 2, 64 * 1536, 1536 * 512, 64 * 512, 1, 0, sk, ws), 4 * 64 * 1536 * 512, 6.3e6)
cs = rnd(64)
t("colsum [R,64]", lambda: L.colsum(x64, 64, R, 64, cs, 1, ws), 0, R * 256)
t("colsum [R,96 ld192]", lambda: L.colsum(o192, 192, R, 96, cs.new_empty(96), 1, ws), 0, R * 384)
stat = rnd(2, 64)
t("bn_stats [R,64]", lambda: L.bn_stats(x64, 64, R, 64, 1e-5, 0.1, stat, cs, cs.clone(), ws), 0, R * 256)
t("bn_apply mish", lambda: L.bn_apply(x64, 64, stat, b64, b64, None, 64, y64, 64, R, 64, 2), 0, R * 512)
t("axpby", lambda: L.axpby(x64, 1.0, y64, 1.0, y64, R * 64, 0), 0, R * 768)
gi, out, sv = rnd(R, 192), rnd(R, 64), rnd(R, 2, 4, 32)
whh, bhh = rnd(96, 32) * 0.2, rnd(96)
t("gru32 fwd horizontal", lambda: L.gru32_fwd(gi, whh, whh, bhh, bhh, out, sv, B * 16, 64, 1, 64, 0, 1), 0, R * (768 + 256 + 1024))
t("gru32 fwd vertical", lambda: L.gru32_fwd(gi, whh, whh, bhh, bhh, out, sv, B * 64, 16, 64, 1024, 1, 64), 0, R * (768 + 256 + 1024))
dgi, dgh, hp = rnd(R, 192), rnd(R, 64), rnd(R, 64)
t("gru32 bwd horizontal", lambda: L.gru32_bwd(x64, out, sv, whh, whh, dgi, dgh, hp, B * 16, 64, 1, 64, 0, 1), 0, R * 3000)
t("gru32 bwd vertical", lambda: L.gru32_bwd(x64, out, sv, whh, whh, dgi, dgh, hp, B * 64, 16, 64, 1024, 1, 64), 0, R * 3000)
dWs, dbs = rnd(64, 64), rnd(64)
t("tn_small dW+db [64,64] K=R", lambda: L.tn_small(x64, 64, y64, 64, dWs, 64, dbs, 64, 64, R, ws), 2 * R * 64 * 64, R * 512)
dW96b, db96 = rnd(96, 64), rnd(96)
t("tn_small dW+db [96,64] ld192", lambda: L.tn_small(o192, 192, y64, 64, dW96b, 64, db96, 96, 64, R, ws), 2 * R * 96 * 64, R * 640)
t("tn_small dW+db [96,32] ld192", lambda: L.tn_small(o192, 192, y64, 64, dW96, 32, db96, 96, 32, R, ws), 2 * R * 96 * 32, R * 512)
for sk in (128, 256):
    t(f"conv3x3 64->64 wgrad sk={sk}", lambda: L.conv_wgrad(x64, y64, dw, B, 16, 64, 64, 64, 3, 3, 64, 64, sk, ws, 1), 2 * R * 64 * 576, R * 512)
for sk in (32, 64):
    t(f"conv3x3 64->256 wgrad sk={sk}", lambda: L.conv_wgrad(x64, y256, dw256, B, 16, 64, 64, 256, 3, 3, 64, 256, sk, ws, 1), 2 * R * 256 * 576, R * 1280)
for sk in (64, 128, 256):
    t(f"conv9x9 64->4 wgrad sk={sk}", lambda: L.conv_wgrad(up, pre, dw994, B, 32, 128, 64, 4, 9, 9, 64, 4, sk, ws, 1), 2 * 4 * R * 4 * 5184, 4 * R * 272)
t("pixshuf_mish fwd", lambda: L.pixshuf_mish_fwd(y256, up, B, 16, 64, 64), 0, R * 2048)

# fused RPE steps
gi_, whf, whb, bhf, bhb = rnd(2, 64, 1536), rnd(1536, 512) * 0.05, rnd(1536, 512) * 0.05, rnd(1536), rnd(1536)
hst, gha, qp_ = torch.zeros(2, B + 1, 64, 512, device="cuda"), rnd(2, B, 64, 1536), rnd(B, 16, 64, 64)
dsh_, dsg_ = (B + 1) * 64 * 512, B * 64 * 1536
t("rpe_step_fwd (fused)", lambda: L.rpe_step_fwd(gi_, whf, whb, bhf, bhb, hst[0, 3], hst[0, 4], gha[0, 3], qp_, 3, B, dsg_, dsh_), 2 * 2 * 64 * 1536 * 512, 0)
whT, car, dga, dgs = rnd(2, 512, 1536), rnd(2, 64, 512), rnd(2, B, 64, 1536), torch.zeros(2, 64, 1536, device="cuda")
t("rpe_step_bwd (fused)", lambda: L.rpe_step_bwd(gi_, gha[0, 3], bhf, bhb, hst[0, 3], qp_, car, dga[0, 4], dga[0, 3], dgs, whT, 3, B, dsg_, dsh_), 2 * 2 * 64 * 1536 * 512, 0)
t("rpe_whh_transpose", lambda: L.rpe_whh_transpose(whf, whb, whT), 0, 2 * 1536 * 512 * 8)

# CRNN prior generator forward (teacher), B = 64
from tatt_amd.model.crnn import CRNN
crnn = CRNN().cuda().eval()
lr_img = torch.rand(B, 4, 16, 64, device="cuda")
with torch.no_grad():
    t("CRNN forward from LR images (B=64)", lambda: crnn.forward_from_images(lr_img), 0, 0, iters=5)
