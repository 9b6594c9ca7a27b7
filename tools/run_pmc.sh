#!/bin/bash
# usage (GPU box, repo root): bash tools/run_pmc.sh <tag>   -- HBM-side traffic of the SRB-shaped conv launch, one counter per run
set -e
tag=$1
out=$PWD/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $out/$c -o p -- python3 $GRAFT_REPO_ROOT/tools/conv_only.py 5 > $out/$c.log 2>&1
  f=$(find $out/$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$c" >> $out/summary.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gemm_kernel" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]]
vals = [float(r["Counter_Value"]) for r in rows]
print(sys.argv[2], "launches", len(vals), "last", vals[-1] if vals else None, "mean", sum(vals) / max(1, len(vals)))
PY
done
cat $out/summary.txt
