#!/bin/bash
# usage (GPU box, repo root): bash tools/run_gaps.sh <tag> [bench args] -- graph-mode kernel trace -> tools/graph_gaps.py + graph_timeline.py
tag=$1; shift
out=$PWD/gpurun_out/gaps_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/g -o g -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $out/graph_bench.log 2>&1
db=$(find $out/g -name "*.db" | head -1)
python3 $GRAFT_REPO_ROOT/tools/graph_gaps.py $db > $out/gaps.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/graph_timeline.py $db 40 > $out/graph_timeline.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/step_dump.py $db > $out/step_dump.txt 2>&1 || true
python3 $GRAFT_REPO_ROOT/tools/alone_time.py $db 45 > $out/alone.txt 2>&1 || true
rm -rf $out/g
