#!/bin/bash
# round 5: decode A/B of the L2 touch-prefetch (MMTG_DECODE_TOUCH) in the bf16 and bf16x3 token steps
mkdir -p gpurun_out/r5d
for dt in bf16x3 bf16; do for t in 0 1; do
echo "== $dt touch=$t" | tee -a gpurun_out/r5d/touch_ab.txt
MMTG_DECODE_TOUCH=$t timeout 600 python3 bench.py --mode decode --dtype $dt --steps 3 --warmup 1 --no-roofline --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['us_per_token_step'], d['check'])" | tee -a gpurun_out/r5d/touch_ab.txt
done; done
timeout 900 python -m pytest tests/test_decode_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "teacher_forced" 2>&1 | tail -5 | tee gpurun_out/r5d/tests_touch0.txt
MMTG_DECODE_TOUCH=1 timeout 900 python -m pytest tests/test_decode_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "teacher_forced or kv_cache" 2>&1 | tail -5 | tee gpurun_out/r5d/tests_touch1.txt
