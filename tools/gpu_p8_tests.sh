#!/bin/bash
mkdir -p gpurun_out/big
timeout 1800 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -6 | tee gpurun_out/big/pytest_ops.txt
bash tools/gpu_ab.sh "MMTG_GEMM_P8=0" "X=0"
python tools/step_breakdown.py 5 2>&1 | head -24 | tee gpurun_out/big/step_breakdown_p8.txt
