#!/bin/bash
mkdir -p gpurun_out/r6m
( time timeout 3300 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider ) 2>&1 | tail -8 | tee gpurun_out/r6m/pytest_gpu.txt
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -6 | tee gpurun_out/r6m/smoke.txt
timeout 1500 python bench.py > gpurun_out/r6m/bench_default.json 2> gpurun_out/r6m/bench_default.err; tail -c 300 gpurun_out/r6m/bench_default.json
