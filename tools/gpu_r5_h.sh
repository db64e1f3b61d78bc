#!/bin/bash
# round 5: split-precision attention -- op test, model parity, timing A/B
mkdir -p gpurun_out/r5h
timeout 900 python -m pytest tests/test_x3_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "attention" 2>&1 | tail -25 | tee gpurun_out/r5h/attn_op.txt
timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_decode_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "x3" 2>&1 | tail -15 | tee gpurun_out/r5h/x3_model.txt
timeout 300 python3 tools/bench_x3.py bf16x3 64 5 2>&1 | tail -20 | tee gpurun_out/r5h/bench_x3.txt
MMTG_X3_ATTN=0 timeout 300 python3 tools/bench_x3.py bf16x3 64 5 2>&1 | head -8 | tee gpurun_out/r5h/bench_x3_f32attn.txt
