#!/bin/bash
# usage (GPU box, repo root): bash tools/run_tsc_profile.sh <tag> [bench flags]  -> eager per-step kernel table of the config-3 step (bench.py --tsc)
set -e
tag=$1; shift
out=$PWD/gpurun_out/tsc_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/e -o e -- python3 $GRAFT_REPO_ROOT/bench.py --tsc --steps 10 --warmup 3 --no-cpu-baseline --no-graph "$@" > $out/eager_bench.json 2> $out/eager.err
csv=$(find $out/e -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $csv 100 > $out/eager_summary.txt 2>&1 || true
rm -rf $out/e
