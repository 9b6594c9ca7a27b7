#!/bin/bash
# usage (GPU box, repo root): bash tools/run_pmc_step.sh <tag> [bench flags]
# HBM-side traffic of EVERY kernel of the eager training step: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes
# (one counter per run, kernels serialised by the collection), mean per launch and per step by kernel name.
set -e
tag=$1; shift
out=$PWD/gpurun_out/pmcstep_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 170 rocprofv3 --pmc $c --output-format csv -d $out/$c -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/$c.log 2>&1
  cp $(find $out/$c -name "*counter_collection.csv" | head -1) $out/$c.csv
  rm -rf $out/$c
done
python3 $GRAFT_REPO_ROOT/tools/pmc_step_table.py $out/FETCH_SIZE.csv $out/WRITE_SIZE.csv > $out/table.txt
rm -f $out/FETCH_SIZE.csv $out/WRITE_SIZE.csv
head -50 $out/table.txt
