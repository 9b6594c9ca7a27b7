#!/bin/bash
# usage (GPU box, repo root): bash tools/run_full_profile.sh <tag>
# eager-mode per-step kernel table of the whole reference iteration (bench.py --full): where the CRNN strands' time goes
set -e
tag=$1
out=$PWD/gpurun_out/full_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/e -o e -- python3 $GRAFT_REPO_ROOT/bench.py --full --steps 10 --warmup 3 --no-cpu-baseline --no-graph > $out/eager_bench.json 2> $out/eager.err
csv=$(find $out/e -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $csv 90 > $out/eager_summary.txt 2>&1 || true
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $csv 120 grid > $out/eager_by_grid.txt 2>&1 || true
rm -rf $out/e
