#!/bin/bash
mkdir -p gpurun_out/r3x2
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_decode_gpu.py -x -q --no-header -p no:cacheprovider -k "decode_gemm or fused or kv_cache" 2>&1 | tail -3
for v in "2,4,1,4" "2,4,1,3" "2,4,1,2" "2,6,1,4" "3,4,1,4" "2,3,1,4" "2,4,1,6" "4,4,1,4"; do
MMTG_DECODE_SPLITS=$v timeout 600 python bench.py --mode decode --no-cpu-baseline --no-roofline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('splits $v', d['value'], d['config']['us_per_token_step'])" | tee -a gpurun_out/r3x2/decode_splits.txt
done
