"""Which kernels of a graph-mode step run ALONE on the GPU, and for how long: python tools/alone_time.py <results.db> [top]
(rocprofv3 --kernel-trace of bench.py; last step).  A kernel that runs beside others hides its inefficiency behind them; one that
runs alone puts every microsecond on the step (the TRI_SSIM launches of the config-3 step were found this way)."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.cursor().execute("select name,start,end,queue_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "adam_k" in r[0]]
segs = [rows[a + 1:b + 1] for a, b in zip(idx[:-1], idx[1:]) if b - a > 200]
seg = segs[-1]


def short(nm):
    nm = nm.replace("(anonymous namespace)::", "").replace("void ", "")
    return (nm[:nm.index("(")] if "(" in nm else nm)[:56]


t0, t1 = seg[0][1], max(r[2] for r in seg)
ev = sorted([(s, 1, i) for i, (_, s, e, _) in enumerate(seg)] + [(e, -1, i) for i, (_, s, e, _) in enumerate(seg)])
live, last = set(), t0
alone = collections.defaultdict(float)
idle = 0.0
for t, d, i in ev:
    if len(live) == 1:
        alone[short(seg[next(iter(live))][0])] += t - last
    elif not live:
        idle += t - last
    if d > 0:
        live.add(i)
    else:
        live.discard(i)
    last = t
tot = collections.defaultdict(float)
cnt = collections.Counter()
for n, s, e, _ in seg:
    tot[short(n)] += e - s
    cnt[short(n)] += 1
print(f"step wall {(t1 - t0) / 1e3:.1f} us; no kernel running {idle / 1e3:.1f} us; exactly one kernel running {sum(alone.values()) / 1e3:.1f} us")
print(f"{'kernel':58s} {'calls':>5s} {'alone us':>9s} {'total us':>9s}")
for n, a in sorted(alone.items(), key=lambda kv: -kv[1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{n:58s} {cnt[n]:5d} {a / 1e3:9.1f} {tot[n] / 1e3:9.1f}")
