#!/bin/bash
mkdir -p gpurun_out/gemm
echo "== medium default" | tee gpurun_out/gemm/wide_ab.txt
DMODEL=1024 TOKENS=16384 python tools/bench_gemm.py 2>&1 | grep -v amdgpu | grep "fwd\|dgrad" | tee -a gpurun_out/gemm/wide_ab.txt
echo "== medium NO_WIDE (flags 32)" | tee -a gpurun_out/gemm/wide_ab.txt
FLAGS=32 DMODEL=1024 TOKENS=16384 python tools/bench_gemm.py 2>&1 | grep -v amdgpu | grep "fwd\|dgrad" | tee -a gpurun_out/gemm/wide_ab.txt
echo "== base default" | tee -a gpurun_out/gemm/wide_ab.txt
python tools/bench_gemm.py 2>&1 | grep -v amdgpu | grep "fwd\|dgrad" | tee -a gpurun_out/gemm/wide_ab.txt
echo "== base NO_WIDE" | tee -a gpurun_out/gemm/wide_ab.txt
FLAGS=32 python tools/bench_gemm.py 2>&1 | grep -v amdgpu | grep "fwd\|dgrad" | tee -a gpurun_out/gemm/wide_ab.txt
