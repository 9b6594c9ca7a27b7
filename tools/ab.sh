#!/bin/bash
# usage (GPU box, repo root): bash tools/ab.sh "ENV1=.. ENV2=.." "ENV=.." ...   -> one bench run per argument (env assignments, "-" = none),
# all in one call = same box; prints ms_per_step mean / median / p90 per variant.  Extra bench flags: AB_FLAGS="--tsc".
mkdir -p gpurun_out/ab
i=0
for v in "$@"; do
  i=$((i+1))
  if [ "$v" = "-" ]; then v=""; fi
  env $v timeout -k 10 200 python bench.py --no-cpu-baseline $AB_FLAGS > gpurun_out/ab/$i.json 2> gpurun_out/ab/$i.err || { echo "variant $i ($v) failed"; tail -5 gpurun_out/ab/$i.err; exit 1; }
  python - "$v" gpurun_out/ab/$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
print(f"{sys.argv[1] or '(default)':40s} {d['ms_per_step']:.3f} {d['ms_per_step_median']:.3f} {d['ms_per_step_p90']:.3f}  loss {d['config'].get('loss')}")
PY
done
