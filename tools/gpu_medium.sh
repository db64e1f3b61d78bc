#!/bin/bash
mkdir -p gpurun_out/med
export MMTG_TEST_REPORT=$(pwd)/gpurun_out/med/test_report.jsonl
rm -f $MMTG_TEST_REPORT
timeout 1500 python -m pytest tests/test_model_gpu.py -k "scaled_stress" -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -25 | tee gpurun_out/med/pytest.txt
cat $MMTG_TEST_REPORT
timeout 900 python bench.py --config medium --steps 8 --warmup 3 2> gpurun_out/med/bench.err > gpurun_out/med/bench_medium.json; cut -c1-2000 gpurun_out/med/bench_medium.json; tail -5 gpurun_out/med/bench.err
