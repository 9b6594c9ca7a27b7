#!/bin/bash
# usage (GPU box, repo root): bash tools/run_micro.sh tools/micro_x.py  -> kernel durations of that script under rocprofv3
set -e
out=$PWD/gpurun_out/micro; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --kernel-trace -d $out -o m -- python3 $GRAFT_REPO_ROOT/$1 2>&1 | grep -v "rocprofv3\|simple_timer" || true
python3 $GRAFT_REPO_ROOT/tools/kstats.py $out
