#!/bin/bash
O=gpurun_out/p28; mkdir -p $O
timeout 1200 python -m pytest tests/test_decode_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "decode or greedy or generate or kv_cache or sample" 2>&1 | tail -5 | tee $O/pytest_decode.txt
for i in 1 2; do
python gpurun_ab/base/bench.py --mode decode --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BASE decode tok/s %.0f  us/step %.1f' % (d['value'], d['roofline']['us_per_token_step_hip_events']))" | tee -a $O/decode_ab.txt
python bench.py --mode decode --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NEW  decode tok/s %.0f  us/step %.1f' % (d['value'], d['roofline']['us_per_token_step_hip_events']))" | tee -a $O/decode_ab.txt
done
