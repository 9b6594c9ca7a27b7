"""Per-kernel count / avg / min (ns) from a rocprofv3 rocpd database: python tools/kstats.py <dir-or-db>"""
import glob
import sqlite3
import sys
p = sys.argv[1]
db = p if p.endswith(".db") else glob.glob(p + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
gz = "d.grid_size_z" if "grid_size_z" in cols else ("d.grid_z" if "grid_z" in cols else "0")
q = f"select s.kernel_name || ' z=' || {gz}, count(*), avg(d.end-d.start), min(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by 1 order by 1"
for r in c.execute(q):
    print(f"{r[0][:70]:70s} {r[1]:5d} avg {r[2] / 1e3:8.1f} us  min {r[3] / 1e3:8.1f} us")
