#!/bin/bash
# round 4, probe 5: persistent decode token step -- parity tests first (bounded), then timing (3 and 2 workgroups per CU, per-launch step)
mkdir -p gpurun_out/p5
O=gpurun_out/p5
export MMTG_TEST_REPORT=$(pwd)/$O/test_report.jsonl
timeout 900 python -m pytest tests/test_decode_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "persistent or teacher or fused_decode_step or full_size_batched" 2>&1 | tail -25 > $O/pytest_persist.txt
cat $O/pytest_persist.txt
for v in "" "MMTG_DECODE_PERSIST_WGS=2" "MMTG_DECODE_PERSIST=0"; do
  echo "=== $v"
  env $v timeout 600 python bench.py --mode decode --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], 'tokens/s', d['config']['us_per_token_step'], 'us/step', d['roofline']['frac'], d['roofline']['launches_per_token_step'], d['check'])
"
done > $O/decode_bench_ab.txt 2>&1
cat $O/decode_bench_ab.txt
