#!/bin/bash
# round 5, first x3 run: op tests, model parity in the bf16x3 mode, train-step timing of the three modes
mkdir -p gpurun_out/r5a
timeout 900 python -m pytest tests/test_x3_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -30 | tee gpurun_out/r5a/x3_ops.txt
timeout 1500 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "bf16x3" 2>&1 | tail -40 | tee gpurun_out/r5a/x3_model.txt
timeout 300 python3 tools/bench_x3.py bf16x3 64 5 2>&1 | tail -20 | tee gpurun_out/r5a/bench_x3.txt
timeout 300 python3 tools/bench_x3.py f32 64 3 2>&1 | tail -20 | tee gpurun_out/r5a/bench_f32.txt
