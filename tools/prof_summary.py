"""Per-training-step kernel table from a rocprofv3 --kernel-trace csv: python tools/prof_summary.py <kernel_trace.csv> [top]
Launches that happen after the last optimiser kernel (bench.py's roofline leg) are excluded; counts/times are per step."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "clip_adam" in r["Kernel_Name"] or "adam" in r["Kernel_Name"].lower()]
steps = len([i for i in adam]) or 1
names = collections.Counter(r["Kernel_Name"] for r in rows if "adam" in r["Kernel_Name"].lower())
per_step_adam = min(names.values()) if names else 1
steps = min(names.values()) if names else 1
rows = rows[: adam[-1] + 1] if adam else rows
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    n = n[: n.index("(")] if "(" in n else n
    if len(sys.argv) > 3 and sys.argv[3] == "grid":          # one line per (kernel, grid): separates the shapes a GEMM template serves
        n = f"{n[:46]} [{int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))},{r['Grid_Size_Y']},{r['Grid_Size_Z']}]"
    a = agg[n]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"steps={steps}  kernel time per step = {tot / steps / 1e3:.3f} ms  launches per step = {sum(v[0] for v in agg.values()) / steps:.0f}")
print(f"{'kernel':58s} {'calls/step':>10s} {'ms/step':>9s} {'avg us':>8s} {'%':>6s}")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{n[:58]:58s} {c / steps:10.1f} {t / steps / 1e3:9.3f} {t / c:8.1f} {100 * t / tot:6.1f}")
