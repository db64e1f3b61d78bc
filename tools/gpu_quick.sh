#!/bin/bash
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "gemm" 2>&1 | tail -3
timeout 600 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | grep -v "s=1)" | tail -8
TNSWEEP=1 timeout 600 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | grep "K=15104 s="
