#!/bin/bash
mkdir -p gpurun_out/big
MODE=ref python tools/micro/p8_check.py 2>&1 | grep -v amdgpu.ids | tail -1
for rows in 256 192 0; do
  MMTG_GEMM_P8_ROWS=$rows MMTG_GEMM_BIG=2 timeout 600 python tools/micro/p8_check.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/big/p8_check.txt
done
for v in "MMTG_GEMM_BIG=2" "MMTG_GEMM_BIG=2 MMTG_GEMM_P8_ROWS=192"; do
  echo "---- $v" | tee -a gpurun_out/big/ab2.txt
  env $v NTSET=1 timeout 300 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/big/ab2.txt
  echo "---- $v HBM-cold" | tee -a gpurun_out/big/ab2.txt
  env $v NTSET=1 COLD=1 timeout 300 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/big/ab2.txt
done
bash tools/gpu_ab.sh "X=0" "MMTG_GEMM_BIG=2"
MMTG_GEMM_BIG=2 python tools/step_breakdown.py 5 2>&1 | head -24 | tee gpurun_out/big/step_breakdown_big2.txt
