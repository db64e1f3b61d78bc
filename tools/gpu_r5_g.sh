#!/bin/bash
# round 5: x3 train step after the projector products / fused plane producers / fp32 few-rows slabs; parity re-check in BOTH fp32-storage modes
mkdir -p gpurun_out/r5g
timeout 2400 python -m pytest tests/test_x3_gpu.py tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "x3 or f32" 2>&1 | tail -15 | tee gpurun_out/r5g/x3_tests.txt
timeout 300 python3 tools/bench_x3.py bf16x3 64 5 2>&1 | tail -20 | tee gpurun_out/r5g/bench_x3.txt
MMTG_NO_FEW_ROWS=1 timeout 300 python3 tools/bench_x3.py bf16x3 64 5 2>&1 | tail -20 | tee gpurun_out/r5g/bench_x3_nofewrows.txt
