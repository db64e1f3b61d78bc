#!/bin/bash
mkdir -p gpurun_out/big
MODE=ref python tools/micro/p8_check.py 2>&1 | grep -v amdgpu.ids | tail -2
MMTG_GEMM_BIG=2 timeout 600 python tools/micro/p8_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/big/p8_check.txt
for v in "X=0" "MMTG_GEMM_BIG=2"; do
  echo "---- $v" | tee -a gpurun_out/big/ab.txt
  env $v NTSET=1 timeout 300 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/big/ab.txt
  echo "---- $v HBM-cold" | tee -a gpurun_out/big/ab.txt
  env $v NTSET=1 COLD=1 timeout 300 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/big/ab.txt
done
