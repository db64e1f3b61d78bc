#!/bin/bash
mkdir -p gpurun_out/big
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "gemm" 2>&1 | tail -8 | tee gpurun_out/big/pytest_m32.txt
MMTG_GEMM_P8=0 MODE=ref python tools/micro/p8_check.py 2>&1 | grep -v amdgpu.ids | tail -1
timeout 600 python tools/micro/p8_check.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a gpurun_out/big/p8_check.txt
NTSET=1 timeout 300 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/big/m32_ntset.txt
NTSET=1 COLD=1 timeout 300 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | sed 's/^/cold /' | tee -a gpurun_out/big/m32_ntset.txt
python tools/gemm_timeline.py 15104 3072 768 NT 0 2>&1 | grep "K loop per K tile\|kernel span\|epilogue"
python tools/gemm_timeline.py 15104 768 3072 NT 0 2>&1 | grep "K loop per K tile\|kernel span\|epilogue"
python tools/gemm_timeline.py 8192 8192 8192 NT 0 2>&1 | grep "K loop per K tile\|kernel span\|epilogue"
bash tools/gpu_ab.sh "X=0"
