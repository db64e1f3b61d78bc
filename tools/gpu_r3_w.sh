#!/bin/bash
mkdir -p gpurun_out/r3w
timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_ops_gpu.py -x -q --no-header -p no:cacheprovider -k "decode or logits_process or kv_cache or generate or sample" 2>&1 | tail -6 | tee gpurun_out/r3w/pytest_decode.txt
for v in "MMTG_DECODE_FUSED=0" "MMTG_DECODE_FUSED=1" "MMTG_DECODE_FUSED=0" "MMTG_DECODE_FUSED=1"; do
env $v timeout 600 python bench.py --mode decode --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('$v', d['value'], d['config']['us_per_token_step'], r['launches_per_token_step'], r['per_category_ms_per_generation'], d['check'])" | tee -a gpurun_out/r3w/decode_fused_ab.txt
done
