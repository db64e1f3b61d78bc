#!/bin/bash
mkdir -p gpurun_out/r3h
for abl in 0 1 2; do echo "=== MMTG_WGRAD_ABLATE=$abl (1: one ds_read_b128 per fragment instead of two transposed reads; 2: no LDS-DMA fills after the first K tile; timing only)"; MMTG_WGRAD_ABLATE=$abl SPLITS=2 SPLITS8=2 timeout 300 python tools/bench_wgrad_group.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r3h/wgrad_group_ablation.txt
timeout 300 python tools/bench_ln_floor.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3h/ln_fwd_floor.txt
timeout 900 python -m pytest tests/test_model_gpu.py -x -q --no-header -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/r3h/pytest_model.txt
