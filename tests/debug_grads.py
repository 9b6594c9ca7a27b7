"""GPU debugging aid: per-parameter gradient errors and a few intermediate gradients vs the CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import procgen, tatt_ref as R
from tests.util import leafify, make_state, rel_l2
from tests.test_gpu_model import build, stages

B, K, seed, p = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 37, 5, 0.0
state = make_state(K, seed=1)
m = build(K, seed=1).train()
hr, lr, tp = procgen.make_inputs(B, K, seed=2)
sr, ctx = m.engine().forward(lr.cuda(), tp.cuda(), True, seed, p_drop=p)
P = leafify(state)
inter = {}
sr_o = R.tsrn_forward(P, lr, tp, training=True, inter=inter)
for k in ("ctrl_points", "tps_out", "block1"):
    inter[k].retain_grad()
st = stages(ctx, B)
print("ctrl", rel_l2(ctx["stn"][5].view(B, 20, 2).cpu().numpy(), inter["ctrl_points"].detach().numpy()))
xs = ctx["xs"].view(B, 16, 64, 4).permute(0, 3, 1, 2)
print("tps_out", rel_l2(xs.cpu().numpy(), inter["tps_out"].detach().numpy()))
for k in ["block1", "query_pos", "memory", "tp_map", "block2", "block4", "block6", "pre_sigmoid"]:
    print(k, rel_l2(st[k].cpu().numpy(), inter[k].detach().numpy()))
print("sr maxabs", float((sr.cpu() - sr_o.detach()).abs().max()))
L = m.engine().L
dsr = torch.empty_like(sr)
L.image_loss_bwd(sr, hr.cuda(), None, 100.0 / B, dsr, B, 4, 32, 128, 1.0, 1e-4)
G = {n: torch.zeros_like(q) for n, q in m.named_parameters()}
eng = m.engine()
# capture dxs / dctrl by wrapping the tps backward call
orig = L.tps_sample_bwd
cap = {}
def wrap(x, ctrl, T, dy, dctrl, *a):
    r = orig(x, ctrl, T, dy, dctrl, *a)
    cap["dxs"], cap["dctrl"] = dy.clone(), dctrl.clone()
    return r
L.tps_sample_bwd = wrap
eng.backward(ctx, dsr, G)
(R.image_loss(sr_o, hr).mean() * 100).backward()
print("dxs", rel_l2(cap["dxs"].view(B, 16, 64, 4).permute(0, 3, 1, 2).cpu().numpy(), inter["tps_out"].grad.numpy()))
print("dctrl", rel_l2(cap["dctrl"].view(B, 20, 2).cpu().numpy(), inter["ctrl_points"].grad.numpy()))
for n in G:
    ref = P[n].grad
    if ref is None:
        continue
    e = rel_l2(G[n].cpu().numpy(), ref.numpy())
    if e > 3e-4:
        print(f"{e:.2e} {n} |ref|={ref.norm().item():.3e}")

# ---- stage-local check of the STN backward: recompute each stage on the CPU (fp64) from the GPU's own inputs ----
import torch.nn.functional as F
from tatt_amd.engine import STN_STAGES
print("---- STN stage-local backward ----")
eng.debug = {}
G2 = {n: torch.zeros_like(q) for n, q in m.named_parameters()}
eng.backward(ctx, dsr, G2)
stn_saved = ctx["stn"][0]
sd = {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
for si, ((idx, cin, cout, pool), rec) in enumerate(zip(STN_STAGES, stn_saved)):
    hin, c, stat, a, hh, ww = rec
    dh_up, da, dc = [t.cpu().double() for t in eng.debug[f"stn{idx}"]]
    pre = f"stn_head.stn_convnet.{idx}."
    x = hin.cpu().double().view(B, hh, ww, cin).permute(0, 3, 1, 2).requires_grad_(True)
    W_ = sd[pre + "0.weight"].clone().requires_grad_(True)
    bb = sd[pre + "0.bias"].clone().requires_grad_(True)
    g_ = sd[pre + "1.weight"].clone().requires_grad_(True)
    be = sd[pre + "1.bias"].clone().requires_grad_(True)
    cc = F.conv2d(x, W_, bb, padding=1); cc.retain_grad()
    aa = torch.relu(F.batch_norm(cc, None, None, g_, be, True, 0.1, 1e-5)); aa.retain_grad()
    yy = F.max_pool2d(aa, pool, pool) if pool else aa
    up = dh_up.view(B, yy.shape[2], yy.shape[3], cout).permute(0, 3, 1, 2)
    yy.backward(up)
    nh = lambda t, C_: t.view(B, hh, ww, C_).permute(0, 3, 1, 2)
    print(idx, "fwd c %.1e a %.1e" % (rel_l2(nh(c.cpu().double(), cout).numpy(), cc.detach().numpy()), rel_l2(nh(a.cpu().double(), cout).numpy(), aa.detach().numpy())),
          "da %.1e dc %.1e" % (rel_l2(nh(da, cout).numpy(), aa.grad.numpy()), rel_l2(nh(dc, cout).numpy(), cc.grad.numpy())),
          "dW %.1e dgamma %.1e" % (rel_l2(G2[pre + "0.weight"].cpu().numpy(), W_.grad.numpy()), rel_l2(G2[pre + "1.weight"].cpu().numpy(), g_.grad.numpy())),
          end=" ")
    if si > 0:
        prev_idx = STN_STAGES[si - 1][0]
        dh_out = eng.debug[f"stn{prev_idx}"][0].cpu().double()
        print("dhin %.1e" % rel_l2(nh(dh_out, cin).numpy(), x.grad.numpy()))
    else:
        print()

print("---- fp64 oracle as referee ----")
P64 = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() and P[k].requires_grad else v.clone().double() if v.is_floating_point() else v.clone()) for k, v in state.items()}
sr64 = R.tsrn_forward(P64, lr.double(), tp.double(), training=True)
(R.image_loss(sr64, hr.double()).mean() * 100).backward()
print("sr: gpu-vs-64 %.2e  oracle32-vs-64 %.2e" % (float((sr.cpu().double() - sr64.detach()).abs().max()), float((sr_o.detach().double() - sr64.detach()).abs().max())))
worst = []
for n in G:
    if P64[n].grad is None:
        continue
    r = P64[n].grad.numpy()
    worst.append((rel_l2(G[n].cpu().double().numpy(), r), rel_l2(P[n].grad.double().numpy(), r), n, float(P64[n].grad.norm())))
worst.sort(reverse=True)
for a_, b_, n, nr in worst[:25]:
    print(f"gpu-vs-64 {a_:.2e}  oracle32-vs-64 {b_:.2e}  |g|={nr:.2e} {n}")
