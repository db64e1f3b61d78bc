#!/bin/bash
# v6 evidence: per-shape tables, split sweep, in-situ breakdown, decode line
mkdir -p gpurun_out
( echo "# python tools/bench_gemm.py   (warm: 20 back-to-back launches per shape)"; timeout 600 python tools/bench_gemm.py 2>&1 | grep -v "amdgpu.ids\|s=1)";
  echo; echo "# COLD=1 python tools/bench_gemm.py   (every launch behind a 1 GiB fill)"; COLD=1 timeout 600 python tools/bench_gemm.py 2>&1 | grep -v "amdgpu.ids\|s=1)";
  echo; echo "# FLAGS=1024 python tools/bench_gemm.py   (2-stage kernels only: MMTG_GEMM_NO_OCC4, warm)"; FLAGS=1024 timeout 600 python tools/bench_gemm.py 2>&1 | grep -v "amdgpu.ids\|s=1)" ) > gpurun_out/gemm_per_shape_v6.log
( echo "# TNSWEEP=1 SWEEPWIDE=1 python tools/bench_gemm.py   (single-stage weight-gradient kernel, warm)"; TNSWEEP=1 SWEEPWIDE=1 timeout 900 python tools/bench_gemm.py 2>&1 | grep "K=15104 s=";
  echo; echo "# COLD=1 TNSWEEP=1 SWEEPWIDE=1 python tools/bench_gemm.py"; COLD=1 TNSWEEP=1 SWEEPWIDE=1 timeout 900 python tools/bench_gemm.py 2>&1 | grep "K=15104 s=" ) > gpurun_out/gemm_tn_split_sweep_v6.log
timeout 900 python tools/step_breakdown.py 5 2>&1 | grep -v amdgpu.ids > gpurun_out/step_breakdown_v6.log
timeout 600 python bench.py --mode decode --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_decode_v6.json
( for v in 0 1 0 1; do if [ $v = 1 ]; then export MMTG_GEMM_NO_OCC4=1; else unset MMTG_GEMM_NO_OCC4; fi; echo "MMTG_GEMM_NO_OCC4=$v python bench.py --no-cpu-baseline --no-roofline"; timeout 900 python bench.py --no-cpu-baseline --no-roofline 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*'; done ) > gpurun_out/occ4_ab_v6.log
head -5 gpurun_out/step_breakdown_v6.log; cut -c1-200 gpurun_out/bench_decode_v6.json; cat gpurun_out/occ4_ab_v6.log
