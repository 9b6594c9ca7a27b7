#!/bin/bash
# usage (GPU box, repo root): bash tools/run_eager_ab.sh <other libtatt_hip.so>  -> eager per-kernel tables of the in-tree library and of another build, same box
set -e
out=$PWD/gpurun_out/eager_ab
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in new old new2 old2; do
  if [ "${v:0:3}" = "old" ]; then export TATT_LIB=$1; else unset TATT_LIB; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/$v -o e -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph > $out/$v.json 2> $out/$v.err
  python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $(find $out/$v -name "*kernel_trace.csv" | head -1) 12 > $out/$v.txt
  rm -rf $out/$v
  echo "== $v"; head -8 $out/$v.txt | cut -c1-40,59-100
done
