#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "attention or model or train or grad" 2>&1 | tail -3
timeout 900 python tools/step_breakdown.py 5 2>&1 | grep -v amdgpu.ids > gpurun_out/step_breakdown_v5.log; head -4 gpurun_out/step_breakdown_v5.log; grep "colsum" gpurun_out/step_breakdown_v5.log
