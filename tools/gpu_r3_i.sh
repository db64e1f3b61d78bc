#!/bin/bash
mkdir -p gpurun_out/r3i
for v in "" 1 2 3; do MMTG_GEMM_TN_BIG=$v CHECK=1 timeout 200 python tools/bench_tn_big.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r3i/tn_big.txt
