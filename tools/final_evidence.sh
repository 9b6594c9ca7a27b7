#!/bin/bash
# usage (GPU box, repo root): bash tools/final_evidence.sh -- all-modes table, final graph + eager profiles, HBM traffic of the bf16-input conv
bash tools/all_modes.sh > gpurun_out/all_modes.txt 2>&1
bash tools/run_final_profile.sh r03c > gpurun_out/prof_r03c.log 2>&1
bash tools/run_pmc_kernel.sh convx16 gemm_halo_kernel tools/conv_only.py 5 10 > gpurun_out/pmc_convx16.log 2>&1
timeout -k 10 200 python bench.py --K 2448 --no-cpu-baseline > gpurun_out/bench_K2448.json 2> gpurun_out/bench_K2448.err
timeout -k 10 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
