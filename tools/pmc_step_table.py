"""Per-kernel HBM traffic table of the eager training step from two rocprofv3 --pmc csv files (FETCH_SIZE, WRITE_SIZE; KB per launch):
python tools/pmc_step_table.py fetch.csv write.csv.  Launches after the last optimiser kernel (bench.py's roofline leg) and before
the first one of the LAST timed steps are dropped by counting optimiser launches: only the last `steps` steps are kept.
Reads are shown at face value and doubled (gfx950 tallies 128-byte requests of wide streaming reads at 64 B: MI355X_MICROARCH.md)."""
import collections
import csv
import sys


def load(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    adam = [i for i, r in enumerate(rows) if "adam_k" in r["Kernel_Name"]]
    steps = 3
    if len(adam) > steps:
        rows = rows[adam[-steps - 1] + 1: adam[-1] + 1]
    else:
        rows, steps = rows[: adam[-1] + 1], max(1, len(adam))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        n = n[: n.index("(")] if "(" in n else n
        wg = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])) if "Grid_Size" in r and r["Grid_Size"] else 0
        a = agg[(n[:52], wg)]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg, steps


f, steps = load(sys.argv[1], "FETCH_SIZE")
w, _ = load(sys.argv[2], "WRITE_SIZE")
keys = sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, [0, 0])[1] + w.get(k, [0, 0])[1]))
tot_f = sum(v[1] for v in f.values()) / steps / 1024
tot_w = sum(v[1] for v in w.values()) / steps / 1024
print(f"steps={steps}  per step: FETCH_SIZE {tot_f:.0f} MB at face value ({2 * tot_f:.0f} MB doubled), WRITE_SIZE {tot_w:.0f} MB")
print(f"{'kernel [workgroups]':64s} {'calls/step':>10s} {'fetch MB':>9s} {'x2':>8s} {'write MB':>9s}   per launch")
for k in keys[:70]:
    c = max(f.get(k, [0, 0])[0], w.get(k, [0, 0])[0])
    fm = f.get(k, [0, 0])[1] / max(1, f.get(k, [1, 0])[0]) / 1024
    wm = w.get(k, [0, 0])[1] / max(1, w.get(k, [1, 0])[0]) / 1024
    print(f"{(k[0] + ' [' + str(k[1]) + ']')[:64]:64s} {c / steps:10.1f} {fm:9.1f} {2 * fm:8.1f} {wm:9.1f}")
