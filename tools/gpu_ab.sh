#!/bin/bash
# same-box A/B of the whole training step: bash tools/gpu_ab.sh "ENV_A=1" "ENV_B=1" ...  (each argument one variant; "" = default)
mkdir -p gpurun_out/ab
i=0
for v in "$@"; do
  for rep in 1 2; do
    env $v python bench.py --no-cpu-baseline --no-decode --no-check --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('%-40s ms/step %.3f  gemm %.3f attn_fwd %.3f attn_bwd %.3f ln %.3f misc %.3f' % ('$v', d['ms_per_step'], r['per_category_ms_per_step']['gemm_bf16'], r['per_category_ms_per_step']['attn_fwd'], r['per_category_ms_per_step']['attn_bwd'], r['per_category_ms_per_step']['layernorm'], r['per_category_ms_per_step']['misc']))
" | tee -a gpurun_out/ab/ab.txt
  done
done
