#!/bin/bash
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export MMTG_GEMM_NO_OCC4=1; else unset MMTG_GEMM_NO_OCC4; fi
  echo "== NO_OCC4=$v"; timeout 900 python bench.py --no-cpu-baseline --no-roofline 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
unset MMTG_GEMM_NO_OCC4
timeout 900 python tools/step_breakdown.py 5 2>&1 | grep "dgelu\| gelu\|s=12\|N=2304 K=768\|instrumented"
FLAGS=512 TNSWEEP=1 SWEEPWIDE=1 timeout 900 python tools/bench_gemm.py 2>&1 | grep "768x768 K=15104 s="
