"""Where a graph-mode step spends its wall time: python tools/graph_gaps.py <results.db>
For each of the last steps of a rocprofv3 --kernel-trace run of bench.py: wall time, time with no kernel running at all (launch gaps /
waits), time with exactly one kernel running, and the step's timeline cut into 20 slices with the kernels that dominate each slice."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.cursor().execute("select name,start,end,queue_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "adam_k" in r[0]]
segs = [rows[a + 1:b + 1] for a, b in zip(idx[:-1], idx[1:]) if b - a > 200][-3:]


def short(nm):
    nm = nm.replace("(anonymous namespace)::", "").replace("void ", "")
    return (nm[:nm.index("(")] if "(" in nm else nm)[:40]


for seg in segs[-1:]:
    t0, t1 = seg[0][1], max(r[2] for r in seg)
    ev = sorted([(s, 1) for _, s, e, _ in seg] + [(e, -1) for _, s, e, _ in seg])
    depth, last, hist = 0, t0, collections.Counter()
    for t, d in ev:
        hist[min(depth, 3)] += t - last
        depth += d
        last = t
    print(f"wall {(t1 - t0) / 1e6:.3f} ms; nothing running {hist[0] / 1e6:.3f} ms, one kernel {hist[1] / 1e6:.3f}, two {hist[2] / 1e6:.3f}, three or more {hist[3] / 1e6:.3f}; launches {len(seg)}")
    n = 24
    for k in range(n):
        a, b = t0 + (t1 - t0) * k // n, t0 + (t1 - t0) * (k + 1) // n
        c = collections.Counter()
        for nm, s, e, q in seg:
            ov = min(e, b) - max(s, a)
            if ov > 0:
                c[short(nm)] += ov
        idle = (b - a) - sum(min(e, b) - max(s, a) for _, s, e, _ in [(0, s, e, 0) for _, s, e, _ in seg] if min(e, b) > max(s, a)) if False else 0
        print(f"  {(a - t0) / 1e6:6.2f}-{(b - t0) / 1e6:5.2f} ms: " + ", ".join(f"{nm} {v / (b - a):.2f}" for nm, v in c.most_common(4)))
