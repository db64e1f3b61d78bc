#!/bin/bash
# full GPU validation: every -m gpu test, smoke, default bench
mkdir -p gpurun_out/full
export MMTG_TEST_REPORT=$(pwd)/gpurun_out/full/test_report.jsonl
rm -f $MMTG_TEST_REPORT
timeout 2400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -${1:-15} | tee gpurun_out/full/pytest_gpu.txt
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/full/smoke.txt
timeout 900 python bench.py 2> gpurun_out/full/bench_default.err > gpurun_out/full/bench_default.json; cut -c1-1500 gpurun_out/full/bench_default.json; tail -3 gpurun_out/full/bench_default.err
