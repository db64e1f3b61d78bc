#!/bin/bash
# round 3, call A: GPU test suite + default bench line on the round's first sources
mkdir -p gpurun_out/r3a
timeout 1500 python -m pytest tests -m gpu -x -q --no-header -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/r3a/pytest_gpu.txt
timeout 900 python bench.py 2> gpurun_out/r3a/bench_default.err | tail -1 > gpurun_out/r3a/bench_default.json; cut -c1-700 gpurun_out/r3a/bench_default.json
tail -3 gpurun_out/r3a/bench_default.err
