#!/bin/bash
# full GPU validation: every -m gpu test, smoke, default bench
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -5
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench_default.json; cat gpurun_out/bench_default.json | cut -c1-1800
