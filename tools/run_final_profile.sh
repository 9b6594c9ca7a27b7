#!/bin/bash
# usage (GPU box, repo root): bash tools/run_final_profile.sh <tag>
# rocprofv3 --kernel-trace --stats of the exact default bench command (graph mode) -> kernel_stats.csv, plus the eager-mode per-step table
set -e
tag=$1
out=$PWD/gpurun_out/final_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/b -o b -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
cp $(find $out/b -name "*kernel_stats.csv" | head -1) $out/bench_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/e -o e -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph > $out/eager_bench.json 2> $out/eager.err
csv=$(find $out/e -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $csv 70 > $out/eager_summary.txt 2>&1 || true
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $csv 80 grid > $out/eager_by_grid.txt 2>&1 || true
rm -rf $out/b $out/e
