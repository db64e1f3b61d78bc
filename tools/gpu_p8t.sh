#!/bin/bash
mkdir -p gpurun_out/big
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "gemm" 2>&1 | tail -6 | tee gpurun_out/big/pytest_p8t.txt
for v in "MMTG_GEMM_P8T=0" "X=0"; do
  for slab in 2 1; do
    echo "---- $v SLAB=$slab (2 = product alone, 1 = + slab sum)" | tee -a gpurun_out/big/tn.txt
    env $v TNSET=1 SLAB=$slab timeout 300 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/big/tn.txt
    env $v TNSET=1 SLAB=$slab COLD=1 timeout 300 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | sed 's/^/cold /' | tee -a gpurun_out/big/tn.txt
  done
done
bash tools/gpu_ab.sh "MMTG_GEMM_P8T=0" "X=0"
