#!/bin/bash
mkdir -p gpurun_out/r3f
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q --no-header -p no:cacheprovider -k "eight_phase or epilogues or rowdot" 2>&1 | tail -3 | tee gpurun_out/r3f/pytest_ops.txt
rm -f gpurun_out/ab/ab.txt
bash tools/gpu_ab.sh "MMTG_GEMM_P8_TOUCH=0" "MMTG_GEMM_P8_TOUCH=1"
cp gpurun_out/ab/ab.txt gpurun_out/r3f/ab_p8_touch.txt
python tools/step_breakdown.py 5 > gpurun_out/r3f/step_breakdown.txt 2>&1
MMTG_GEMM_P8_TOUCH=0 python tools/step_breakdown.py 5 > gpurun_out/r3f/step_breakdown_notouch.txt 2>&1
timeout 1200 python -m pytest tests/test_model_gpu.py -x -q --no-header -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/r3f/pytest_model.txt
