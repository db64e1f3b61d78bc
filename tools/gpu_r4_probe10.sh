#!/bin/bash
mkdir -p gpurun_out/p10
timeout 900 python tools/bf16x3_projection.py 2>&1 | grep -v amdgpu > gpurun_out/p10/bf16x3_projection.txt; cat gpurun_out/p10/bf16x3_projection.txt
