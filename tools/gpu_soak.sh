#!/bin/bash
mkdir -p gpurun_out/soak
FULL=1 STAGE=3 TRIALS=12 timeout 1200 python tools/determinism_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/soak/determinism_full_size.txt
timeout 900 python tools/train_curve.py 3000 2>&1 | grep -v amdgpu.ids | tail -25 | tee gpurun_out/soak/train_soak_3000_steps.txt
