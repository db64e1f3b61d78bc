#!/bin/bash
# Round 6: the dGELU product (gemm_occ4 by default: 0.24 of the matrix peak) on the eight-phase kernel, with and without the forward
# storing gelu'(u) instead of u -- whole training step, same box, alternating.
mkdir -p gpurun_out/r6k
for rep in 1 2; do
for cfg in "0 1" "0 2" "1 1" "1 2"; do
  set -- $cfg
  MMTG_GELU_GRAD=$1 MMTG_GEMM_P8=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --primary-only 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    print('GELU_GRAD=$1 GEMM_P8=$2', d['value'], 'tok/s', d['ms_per_step'], 'ms/step  GEMM family', d['roofline']['frac'], d['roofline']['per_category_ms_per_step']['gemm_bf16'])
" | tee -a gpurun_out/r6k/dgelu_p8_ab.txt
done
done
