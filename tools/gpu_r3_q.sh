#!/bin/bash
mkdir -p gpurun_out/r3q
for v in "MMTG_WGRAD_GROUP=0" "MMTG_WGRAD_GROUP=1" "MMTG_WGRAD_GROUP_SPLITS=2" "MMTG_WGRAD_GROUP_CFG=1" "MMTG_WGRAD_GROUP_CFG=1 MMTG_WGRAD_GROUP_SPLITS=2"; do
  env $v timeout 600 python bench.py --config medium --steps 8 --warmup 3 --no-decode --no-cpu-baseline --no-check 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-55s ms/step %.3f tok/s %.0f gemm %.3f (frac %.3f) attn_fwd %.3f attn_bwd %.3f ln %.3f misc %.3f' % ('$v', d['ms_per_step'], d['value'], r['per_category_ms_per_step']['gemm_bf16'], r['frac'], r['per_category_ms_per_step']['attn_fwd'], r['per_category_ms_per_step']['attn_bwd'], r['per_category_ms_per_step']['layernorm'], r['per_category_ms_per_step']['misc']))" | tee -a gpurun_out/r3q/medium_ab.txt
done
