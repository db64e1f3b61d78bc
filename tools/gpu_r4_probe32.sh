#!/bin/bash
O=gpurun_out/p32; mkdir -p $O
timeout 1200 python -m pytest tests/test_decode_gpu.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -4 | tee $O/pytest_decode.txt
( MMTG_DECODE_CHAIN=1 timeout 300 python tools/decode_chain_timeline.py ) 2>&1 | grep -v amdgpu | tee $O/chain_timeline.txt
for v in 0 1; do
MMTG_DECODE_CHAIN=$v timeout 600 python bench.py --mode decode --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MMTG_DECODE_CHAIN=$v decode tok/s %.0f  us/step %.1f  launches %.1f' % (d['value'], d['roofline']['us_per_token_step_hip_events'], d['roofline']['launches_per_token_step']))" | tee -a $O/decode_chain_ab.txt
done
