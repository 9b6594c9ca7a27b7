#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/run_profile.sh <tag> [bench args]
# graph-mode kernel trace -> per-step timeline (tools/graph_timeline.py); eager-mode trace -> per-kernel table (tools/prof_summary.py)
set -e
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/g -o g -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $out/graph_bench.log 2>&1
db=$(find $out/g -name "*.db" | head -1)
python3 $GRAFT_REPO_ROOT/tools/graph_timeline.py $db 45 > $out/graph_timeline.txt 2>&1 || true
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/e -o e -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph "$@" > $out/eager_bench.log 2>&1
csv=$(find $out/e -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $csv 60 > $out/eager_summary.txt 2>&1 || true
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $csv 70 grid > $out/eager_by_grid.txt 2>&1 || true
cp $(find $out/e -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv 2>/dev/null || true
python3 $GRAFT_REPO_ROOT/tools/step_dump.py $db > $out/step_dump.txt 2>&1 || true
rm -rf $out/g $out/e
