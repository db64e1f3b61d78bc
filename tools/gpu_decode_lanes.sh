#!/bin/bash
# same-box A/B of the decode token step: lanes (row blocks side by side) x small-M GEMM ring depth x K splits
mkdir -p gpurun_out/dec
timeout 1200 python -m pytest tests/test_decode_gpu.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -8 | tee gpurun_out/dec/pytest_lanes.txt
for v in "MMTG_DECODE_LANES=1" "MMTG_DECODE_LANES=2" "MMTG_DECODE_LANES=4" "MMTG_DECODE_LANES=8" \
         "MMTG_DECODE_LANES=1 MMTG_SKINNY_CFG=6" "MMTG_DECODE_LANES=1 MMTG_SKINNY_CFG=7" \
         "MMTG_DECODE_LANES=2 MMTG_SKINNY_CFG=6" "MMTG_DECODE_LANES=2 MMTG_SKINNY_CFG=7" "MMTG_DECODE_LANES=4 MMTG_SKINNY_CFG=6" \
         "MMTG_DECODE_LANES=2 MMTG_DECODE_SPLITS=3,4,1,8" "MMTG_DECODE_LANES=2 MMTG_DECODE_SPLITS=2,4,2,8" "MMTG_DECODE_LANES=4 MMTG_DECODE_SPLITS=3,6,2,12" \
         "MMTG_DECODE_LANES=4 MMTG_DECODE_SPLITS=1,2,1,4"; do
  env $v timeout 600 python bench.py --mode decode --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-60s %9.0f tok/s %7.1f us/step' % ('$v', d['value'], d['config']['us_per_token_step']))
" | tee -a gpurun_out/dec/lanes_ab.txt
done
