#!/bin/bash
mkdir -p gpurun_out/r6h
timeout 300 python tools/decode_begin_breakdown.py bf16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6h/decode_begin_breakdown_bf16.txt
timeout 300 python tools/decode_begin_breakdown.py bf16x3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6h/decode_begin_breakdown_bf16x3.txt
timeout 300 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "differentiates_the_forward" 2>&1 | tail -3
