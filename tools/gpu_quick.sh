#!/bin/bash
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "gemm_layouts or weight_gradient" 2>&1 | tail -3
echo "== 256x128 weight-gradient kernel (flags=512), cold"; COLD=1 FLAGS=512 TNSWEEP=1 SWEEPWIDE=1 timeout 900 python tools/bench_gemm.py 2>&1 | grep "K=15104 s="
echo "== 128x128 (flags=0), cold"; COLD=1 TNSWEEP=1 timeout 900 python tools/bench_gemm.py 2>&1 | grep "K=15104 s="
