"""One training step of a graph-mode rocprofv3 trace (rocpd .db) as a time-ordered list: start offset, duration, queue, kernel, and the
number of kernels in flight at its start.  python tools/step_dump.py <results.db>"""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
rows = db.cursor().execute("select name,start,end,queue_id,grid_x,workgroup_x from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "adam_k" in r[0]]
pairs = [(a, b) for a, b in zip(idx[:-1], idx[1:]) if b - a > 300]
a, b = pairs[-2]
seg = rows[a + 1:b + 1]
t0 = seg[0][1]
ends = []
for nm, s, e, q, gx, wx in seg:
    nm = nm.replace("(anonymous namespace)::", "").replace("void ", "")
    nm = nm[:nm.index("(")] if "(" in nm else nm
    ends = [x for x in ends if x > s]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} q{q} inflight={len(ends)} {nm[:60]} [{gx // max(1, wx)}]")
    ends.append(e)
