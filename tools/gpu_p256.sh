#!/bin/bash
mkdir -p gpurun_out/gemm
timeout 900 python -m pytest tests/test_ops_gpu.py -k "persistent_pipeline" -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -12 | tee gpurun_out/gemm/p256_pytest.txt
echo "== default" | tee gpurun_out/gemm/p256_ab.txt
python tools/bench_gemm.py 2>&1 | grep "fwd\|dgrad" | tee -a gpurun_out/gemm/p256_ab.txt
echo "== P256 (flags 4096)" | tee -a gpurun_out/gemm/p256_ab.txt
FLAGS=4096 python tools/bench_gemm.py 2>&1 | grep "fwd\|dgrad" | tee -a gpurun_out/gemm/p256_ab.txt
echo "== cold default" | tee -a gpurun_out/gemm/p256_ab.txt
COLD=1 python tools/bench_gemm.py 2>&1 | grep "fwd\|dgrad" | tee -a gpurun_out/gemm/p256_ab.txt
echo "== cold P256" | tee -a gpurun_out/gemm/p256_ab.txt
COLD=1 FLAGS=4096 python tools/bench_gemm.py 2>&1 | grep "fwd\|dgrad" | tee -a gpurun_out/gemm/p256_ab.txt
