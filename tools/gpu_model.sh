#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider "$@" > gpurun_out/model.log 2>&1
echo "exit $?" >> gpurun_out/model.log
tail -80 gpurun_out/model.log
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -5 gpurun_out/smoke.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_first.log 2>&1; echo "bench exit $?" >> gpurun_out/bench_first.log; tail -12 gpurun_out/bench_first.log
