#!/bin/bash
# round 5: x3 decode step -- parity + timing
mkdir -p gpurun_out/r5b
export MMTG_TEST_REPORT=$(pwd)/gpurun_out/r5b/test_report.jsonl
rm -f $MMTG_TEST_REPORT
timeout 1500 python -m pytest tests/test_decode_gpu.py tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "bf16x3" 2>&1 | tail -40 | tee gpurun_out/r5b/x3_decode_tests.txt
cat $MMTG_TEST_REPORT
for sp in "2,4,1,8" "2,4,1,4" "3,4,1,8" "2,2,1,4" "1,2,1,4"; do
MMTG_DECODE_SPLITS=$sp timeout 600 python3 bench.py --mode decode --dtype bf16x3 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline 2>&1 | tail -2 | cut -c1-600 | tee -a gpurun_out/r5b/decode_x3_splits.txt
done
