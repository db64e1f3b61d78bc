#!/bin/bash
O=gpurun_out/p29; mkdir -p $O
timeout 900 python -m pytest tests/test_decode_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -5 | tee $O/pytest_decode_chain.txt
for i in 1 2; do
for v in 0 1; do
MMTG_DECODE_CHAIN=$v timeout 600 python bench.py --mode decode --no-cpu-baseline 2>$O/err_$v.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('CHAIN=$v decode tok/s %.0f  us/step %.1f  launches %.1f check %s' % (d['value'], d['roofline']['us_per_token_step_hip_events'], d['roofline']['launches_per_token_step'], str(d.get('check'))[:80]))" | tee -a $O/decode_ab.txt
done; done
tail -3 $O/err_1.txt
