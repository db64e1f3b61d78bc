#!/bin/bash
mkdir -p gpurun_out/det
for v in "X=0" "MMTG_GEMM_P8=0" "MMTG_NO_WTE_T=1"; do
  echo "---- $v" | tee -a gpurun_out/det/det.txt
  env $v timeout 600 python tools/determinism_probe.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/det/det.txt
done
