#!/bin/bash
mkdir -p gpurun_out/dec
timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -k "decode or greedy" -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -8 | tee gpurun_out/dec/pytest.txt
for v in "X=1"; do
  env $v python bench.py --mode decode --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', d['value'], 'tok/s', d['config']['us_per_token_step'], 'us/step', d['roofline']['frac'], d['roofline']['per_category_ms_per_generation'])
" | tee -a gpurun_out/dec/ab.txt
done
