#!/bin/bash
mkdir -p gpurun_out/big
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "gemm" 2>&1 | tail -6 | tee gpurun_out/big/pytest_p8p.txt
MMTG_GEMM_P8=0 MODE=ref python tools/micro/p8_check.py 2>&1 | grep -v amdgpu.ids | tail -1
timeout 600 python tools/micro/p8_check.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a gpurun_out/big/p8_check.txt
for v in "MMTG_GEMM_P8_PERSIST=0" "X=0"; do
  echo "---- $v" | tee -a gpurun_out/big/ab3.txt
  env $v NTSET=1 timeout 300 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/big/ab3.txt
  env $v NTSET=1 COLD=1 timeout 300 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | sed 's/^/cold /' | tee -a gpurun_out/big/ab3.txt
done
bash tools/gpu_ab.sh "MMTG_GEMM_P8_PERSIST=0 MMTG_NO_WTE_T=1" "MMTG_GEMM_P8_PERSIST=0" "X=0"
python tools/step_breakdown.py 5 2>&1 | head -30 | tee gpurun_out/big/step_breakdown_p8p.txt
