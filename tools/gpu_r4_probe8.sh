#!/bin/bash
mkdir -p gpurun_out/p8
O=gpurun_out/p8
for v in "" "MMTG_DECODE_KV_NT=1" "" "MMTG_DECODE_KV_NT=1"; do
  echo "=== $v"
  env $v timeout 600 python bench.py --mode decode --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], 'tokens/s', d['config']['us_per_token_step'], 'us/step', d['roofline']['frac'], d['roofline']['per_category_ms_per_generation'])
"
done > $O/decode_kv_nt_ab.txt 2>&1
cat $O/decode_kv_nt_ab.txt
timeout 900 python -m pytest tests/test_decode_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -4 > $O/pytest_decode.txt; cat $O/pytest_decode.txt
