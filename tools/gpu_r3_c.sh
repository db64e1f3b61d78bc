#!/bin/bash
mkdir -p gpurun_out/r3c
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q --no-header -p no:cacheprovider -k "wgrad_group" 2>&1 | tail -3 | tee gpurun_out/r3c/pytest_ops.txt
{ timeout 300 python tools/bench_wgrad_group.py; MMTG_WGRAD_FENCE=1 timeout 300 python tools/bench_wgrad_group.py; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3c/wgrad_group_isolated.txt
rm -f gpurun_out/ab/ab.txt
bash tools/gpu_ab.sh "MMTG_WGRAD_GROUP=0" "MMTG_WGRAD_GROUP=1" "MMTG_WGRAD_GROUP_SPLITS=3"
cp gpurun_out/ab/ab.txt gpurun_out/r3c/ab_wgrad_group.txt
