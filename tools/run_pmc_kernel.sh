#!/bin/bash
# usage (GPU box, repo root): bash tools/run_pmc_kernel.sh <tag> <kernel name substring> <python script> [args...]
# HBM-side traffic of one kernel: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (one counter per run), per launch.
set -e
tag=$1; pat=$2; shift; shift
out=$PWD/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $out/$c -o p -- python3 $GRAFT_REPO_ROOT/"$@" > $out/$c.log 2>&1
  f=$(find $out/$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$c" "$pat" >> $out/summary.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[3] in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]]
vals = [float(r["Counter_Value"]) for r in rows]
print(sys.argv[3], sys.argv[2], "launches", len(vals), "last", vals[-1] if vals else None, "mean", sum(vals) / max(1, len(vals)))
PY
  rm -rf $out/$c
done
cat $out/summary.txt
