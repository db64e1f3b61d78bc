#!/bin/bash
mkdir -p gpurun_out/r3u
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q --no-header -p no:cacheprovider -k "layernorm" 2>&1 | tail -3 | tee gpurun_out/r3u/pytest_ln.txt
for v in "MMTG_LN_NO1024=1" "MMTG_X=0" "MMTG_LN_CAP=768"; do
  env $v timeout 600 python bench.py --config medium --steps 8 --warmup 3 --no-decode --no-cpu-baseline --no-check 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-22s ms/step %.3f tok/s %.0f gemm %.3f (frac %.3f) attn_fwd %.3f attn_bwd %.3f ln %.3f misc %.3f' % ('$v', d['ms_per_step'], d['value'], r['per_category_ms_per_step']['gemm_bf16'], r['frac'], r['per_category_ms_per_step']['attn_fwd'], r['per_category_ms_per_step']['attn_bwd'], r['per_category_ms_per_step']['layernorm'], r['per_category_ms_per_step']['misc']))" | tee -a gpurun_out/r3u/medium_ln_ab.txt
done
