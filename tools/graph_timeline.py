"""Per-step view of a rocprofv3 --kernel-trace run of bench.py in graph mode (rocpd .db): python tools/graph_timeline.py <results.db> [top]
Prints wall / busy / overlapped time per step and the kernels by total time, with their HIP queue."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.cursor().execute("select name,start,end,queue_id,grid_x,grid_y,grid_z,workgroup_x from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "adam_k" in r[0]]
segs = [rows[a + 1:b + 1] for a, b in zip(idx[:-1], idx[1:]) if b - a > 300][-5:]
n = len(segs)
agg = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
wall = busy = par = 0.0
for seg in segs:
    wall += seg[-1][2] - seg[0][1]
    ev = sorted([(s, 1) for _, s, e, *_ in seg] + [(e, -1) for _, s, e, *_ in seg])
    depth, last = 0, None
    for t, d in ev:
        if depth > 0:
            busy += t - last
        if depth > 1:
            par += t - last
        depth += d
        last = t
    for nm, s, e, q, gx, gy, gz, wx in seg:
        nm = nm.replace("(anonymous namespace)::", "").replace("void ", "")
        nm = nm[:nm.index("(")] if "(" in nm else nm
        a = agg[nm]
        a[0] += 1
        a[1] += e - s
        a[2][q] += 1
print(f"steps={n} wall={wall / n / 1e6:.3f} ms busy={busy / n / 1e6:.3f} ms overlapped={par / n / 1e6:.3f} ms kernels/step={sum(a[0] for a in agg.values()) / n:.0f}")
for nm, (c, t, qs) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{nm[:46]:46s} {c / n:6.1f} {t / n / 1e6:7.3f} ms avg {t / c / 1e3:6.1f} us  queues {dict(qs)}")
