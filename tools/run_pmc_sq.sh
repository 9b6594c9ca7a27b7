#!/bin/bash
# usage (GPU box, repo root): bash tools/run_pmc_sq.sh <tag> <flags>   -- SQ counters of the SRB-shaped conv launch (flags 0 / 2 / 4), two passes of <= 8 counters
tag=$1; fl=$2
out=$PWD/gpurun_out/pmcsq_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -o p -- python3 $GRAFT_REPO_ROOT/tools/conv_only.py 5 $fl > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; exit 1; }
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $out/summary.txt <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gemm" in r["Kernel_Name"]]
last = collections.OrderedDict()
for r in rows:
    last[r["Counter_Name"]] = (r["Kernel_Name"].split("(")[0][-60:], float(r["Counter_Value"]))
for k, (n, v) in last.items():
    print(f"{k:28s} {v:16.0f}   {n}")
PY
  t=$(find $out/p$i -name "*kernel_trace.csv" | head -1)
  python3 - "$t" >> $out/summary.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gemm" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("kernel durations us:", " ".join(f"{x:.1f}" for x in d))
PY
  rm -rf $out/p$i
done
cat $out/summary.txt
