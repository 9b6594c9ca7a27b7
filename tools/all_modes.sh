#!/bin/bash
# usage (GPU box, repo root): bash tools/all_modes.sh  -> ms/step of every bench.py mode on one box (README / DESIGN tables)
for fl in "" "--tsc" "--full" "--full --tsc" "--f32x3" "--f32x3 --tsc" "--f32x3 --full" "--f32x3 --full --tsc" "--bf16" "--bf16 --tsc" "--bf16 --full --tsc"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline $fl > gpurun_out/am.json 2> gpurun_out/am.err || { echo "FAILED: $fl"; tail -3 gpurun_out/am.err; exit 1; }
  python - "$fl" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/am.json").read().strip().split("\n")[-1])
print(f"{sys.argv[1] or '(default)':28s} {d['ms_per_step']:.3f} ms  median {d['ms_per_step_median']:.3f}  {d['value']:.0f} img/s  roofline launch_us {d['roofline'].get('launch_us')} frac {d['roofline'].get('frac')}", flush=True)
PY
done
