#!/bin/bash
mkdir -p gpurun_out/big
export MMTG_GEMM_BIG=2
{
python tools/gemm_timeline.py 15104 3072 768 NT 0
python tools/gemm_timeline.py 15104 3072 768 NT 1
python tools/gemm_timeline.py 15104 768 3072 NT 3
python tools/gemm_timeline.py 15104 768 3072 NT 0
python tools/gemm_timeline.py 15104 13440 768 NT 0
python tools/gemm_timeline.py 8192 8192 8192 NT 0
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/big/p8_timeline.txt
