"""Run the B=64 forward+backward pass several times in one process and report every saved tensor / gradient that is not bitwise
equal between runs (no kernel of the library sums in a run-dependent order: the last one that did, tatt_rotate_bwd, is a gather now).
usage: python tools/probe_determinism.py [fp32|f32x3|bf16] [runs]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import procgen  # noqa: E402
from tatt_amd.model.tsrn import TSRN_TL_TRANS  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B, K = 64, 37
m = TSRN_TL_TRANS(text_embedding=K)
m.load_state_dict(procgen.fill_state(m.state_dict(), seed=0), strict=True)
m = m.cuda().train()
m.dropout_p = 0.0
m.matmul_precision = prec
hr, lr, tp = [t.cuda() for t in procgen.make_inputs(B, K, seed=0)]
eng = m.engine()
sd0 = {k: v.clone() for k, v in m.state_dict().items()}


def flat(ctx):
    out = {}

    def walk(pre, v):
        if torch.is_tensor(v):
            out[pre] = v.clone()
        elif isinstance(v, (list, tuple)):
            for i, x in enumerate(v):
                walk(f"{pre}.{i}", x)
    for k, v in ctx.t.items():
        if k not in ("wpack",):
            walk(k, v)
    return out


ref = None
for r in range(runs):
    m.load_state_dict(sd0)
    sr, ctx = eng.forward(lr, tp, True, 0, p_drop=0.0)
    dsr = torch.empty_like(sr)
    eng.L.image_loss_bwd(sr, hr, None, 100.0 / B, dsr, B, 4, 32, 128, 1.0, 1e-4)
    G = {n: torch.zeros_like(q) for n, q in m.named_parameters()}
    eng.backward(ctx, dsr, G)
    torch.cuda.synchronize()
    cur = flat(ctx)
    cur.update({"grad:" + n: g.clone() for n, g in G.items()})
    cur.update({"dbg:" + n: g.clone() for n, g in getattr(eng, "_dbg", {}).items()})   # (tensors an experiment parks in eng._dbg)
    if ref is None:
        ref = cur
        continue
    bad = [(k, float((cur[k].float() - ref[k].float()).abs().max())) for k in ref
           if k in cur and cur[k].shape == ref[k].shape and not torch.equal(cur[k], ref[k])]
    print(f"run {r}: {len(bad)} tensors differ from run 0")
    for k, d in bad[:int(os.environ.get('SHOW', 40))]:
        print(f"   {k}: max|diff| {d:.3e}")


# ---- second part (argument "trace" as third parameter): checksum every tensor argument after every library call, on the call's own
# stream, and report the first calls whose checksums differ between two passes
if len(sys.argv) > 3 and sys.argv[3] == "trace":
    L = eng.L
    trace = []

    def wrap(name, orig):
        def traced(*args):
            rc = orig(*args)
            for i, a in enumerate(args):
                if torch.is_tensor(a) and a.is_cuda and a.dtype == torch.float32 and a.numel() > 0:
                    trace.append((name, i, tuple(a.shape), a.contiguous().view(-1).view(torch.int32).sum(dtype=torch.int64)))
            return rc
        return traced
    for full in L.decls:
        short = full[len("tatt_"):]
        setattr(L, short, wrap(short, getattr(L, short)))
    traces = []
    for r in range(3):
        trace.clear()
        m.load_state_dict(sd0)
        sr, ctx = eng.forward(lr, tp, True, 0, p_drop=0.0)
        dsr = torch.empty_like(sr)
        eng.L.image_loss_bwd(sr, hr, None, 100.0 / B, dsr, B, 4, 32, 128, 1.0, 1e-4)
        G = {n: torch.zeros_like(q) for n, q in m.named_parameters()}
        eng.backward(ctx, dsr, G)
        torch.cuda.synchronize()
        traces.append([(n, i, s, int(v.item())) for n, i, s, v in trace])
    for r in (1, 2):
        a, b = traces[0], traces[r]
        print(f"trace run {r}: {len(a)} / {len(b)} records")
        shown = 0
        for k, (x, y) in enumerate(zip(a, b)):
            if x[:3] != y[:3]:
                print("   call sequence differs at", k, x[:3], y[:3])
                break
            if x[3] != y[3]:
                print(f"   #{k} {x[0]} arg {x[1]} shape {x[2]}: checksum differs")
                shown += 1
                if shown > 25:
                    break

if os.environ.get("PATTERN"):
    for k in os.environ["PATTERN"].split(","):
        a, b = ref[k].float(), cur[k].float()
        d = (a != b)
        print(k, tuple(a.shape), "differing elements", int(d.sum()), "max", float((a - b).abs().max()))
        if d.dim() == 2 and int(d.sum()):
            rows = d.any(1).nonzero().flatten()
            cols = d.any(0).nonzero().flatten()
            print("   rows differing:", rows.numel(), "first", rows[:24].tolist())
            print("   row % 128 histogram (top):", torch.bincount(rows % 128, minlength=128).topk(8))
            print("   row // 128 distinct:", torch.unique(rows // 128).numel(), torch.unique(rows // 128)[:32].tolist())
            print("   cols differing:", cols.tolist())
            r0 = int(rows[0])
            print("   row", r0, "a", a[r0, :8].tolist(), "b", b[r0, :8].tolist())
