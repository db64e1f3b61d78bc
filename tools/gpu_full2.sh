#!/bin/bash
# full -m gpu suite + smoke + default bench + medium bench A/B (eight-phase kernel on / off)
mkdir -p gpurun_out/full
export MMTG_TEST_REPORT=$(pwd)/gpurun_out/full/test_report.jsonl
rm -f $MMTG_TEST_REPORT
timeout 2700 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -12 | tee gpurun_out/full/pytest_gpu.txt
unset MMTG_TEST_REPORT
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/full/smoke.txt
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/full/bench_default.json; cut -c1-700 gpurun_out/full/bench_default.json
for v in "MMTG_GEMM_P8=0" "X=0"; do
  env $v timeout 900 python bench.py --config medium --steps 8 --warmup 3 --no-cpu-baseline --no-decode 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('medium $v', d['ms_per_step'], 'ms/step', d['value'], 'tok/s', d['roofline']['frac'])
" | tee -a gpurun_out/full/medium_ab.txt
done
