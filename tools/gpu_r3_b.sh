#!/bin/bash
# round 3, call B: grouped weight gradients -- op tests, model-level parity, same-box A/B of the whole step
mkdir -p gpurun_out/r3b
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q --no-header -p no:cacheprovider -k "wgrad_group or weight_gradient" 2>&1 | tail -5 | tee gpurun_out/r3b/pytest_ops.txt
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_ddp_gpu.py -x -q --no-header -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/r3b/pytest_model.txt
rm -f gpurun_out/ab/ab.txt
bash tools/gpu_ab.sh "MMTG_WGRAD_GROUP=0" "MMTG_WGRAD_GROUP=1" "MMTG_WGRAD_GROUP_SPLITS=3" "MMTG_WGRAD_GROUP_SPLITS=4" "MMTG_WGRAD_GROUP_SPLITS=1"
cp gpurun_out/ab/ab.txt gpurun_out/r3b/ab_wgrad_group.txt
