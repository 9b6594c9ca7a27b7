"""Per-kernel count / avg / min (us) from a rocprofv3 rocpd database, one line per (kernel, grid): python tools/kstats.py <dir-or-db>"""
import glob
import sqlite3
import sys
p = sys.argv[1]
db = p if p.endswith(".db") else glob.glob(p + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name, grid_x, grid_y, grid_z, workgroup_x, count(*), avg(end-start), min(end-start) from kernels "
                 "group by 1,2,3,4,5 order by 1,2,3,4").fetchall()
for nm, gx, gy, gz, wx, n, avg, mn in rows:
    nm = nm.replace("(anonymous namespace)::", "").replace("void ", "")
    nm = nm[:nm.index("(")] if "(" in nm else nm
    print(f"{nm[:56]:56s} [{gx // max(1, wx)},{gy},{gz}] {n:5d} avg {avg / 1e3:8.1f} us  min {mn / 1e3:8.1f} us")
