#!/bin/bash
# scratch: 256x128 tiles with a 3-deep ring, warm and HBM-cold
mkdir -p gpurun_out
echo "=== warm 256x128x3 (flags=512)"; FLAGS=512 timeout 600 python tools/bench_gemm.py 2>&1 | grep -v "amdgpu.ids\|s=1)"
echo "=== cold 256x128x3 (flags=512)"; COLD=1 FLAGS=512 timeout 600 python tools/bench_gemm.py 2>&1 | grep -v "amdgpu.ids\|s=1)"
