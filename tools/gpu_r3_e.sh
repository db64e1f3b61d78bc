#!/bin/bash
mkdir -p gpurun_out/r3e
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q --no-header -p no:cacheprovider -k "wgrad_group" 2>&1 | tail -3 | tee gpurun_out/r3e/pytest_ops.txt
timeout 300 python tools/bench_wgrad_group.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3e/wgrad_group_isolated.txt
rm -f gpurun_out/ab/ab.txt
bash tools/gpu_ab.sh "MMTG_WGRAD_GROUP_CFG=0" "MMTG_WGRAD_GROUP_CFG=1" "MMTG_WGRAD_GROUP_CFG=1 MMTG_WGRAD_GROUP_SPLITS=3"
cp gpurun_out/ab/ab.txt gpurun_out/r3e/ab_wgrad_group_cfg.txt
MMTG_WGRAD_GROUP_CFG=1 timeout 900 python -m pytest tests/test_model_gpu.py -x -q --no-header -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/r3e/pytest_model_cfg1.txt
