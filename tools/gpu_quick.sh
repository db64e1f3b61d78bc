#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "colsum or layernorm or gemm_epilogues or weight_gradient" 2>&1 | tail -3
timeout 900 python tools/step_breakdown.py 5 2>&1 | grep -v amdgpu.ids > gpurun_out/step_breakdown_v5.log; head -3 gpurun_out/step_breakdown_v5.log; grep "colsum\|dgelu" gpurun_out/step_breakdown_v5.log
TNSWEEP=1 timeout 600 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids > gpurun_out/gemm_tn_split_sweep.log; tail -12 gpurun_out/gemm_tn_split_sweep.log
timeout 900 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
