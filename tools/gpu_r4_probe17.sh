#!/bin/bash
# attention kernels: working tree vs the base build, isolated + tests
O=gpurun_out/p17; mkdir -p $O
mkdir -p gpurun_ab/base/tools; cp tools/bench_attn.py tools/bench_attn_rounds.py gpurun_ab/base/tools/ 2>/dev/null
( echo "== base"; python gpurun_ab/base/tools/bench_attn.py; echo "== new"; python tools/bench_attn.py
  echo "== base"; python gpurun_ab/base/tools/bench_attn.py; echo "== new"; python tools/bench_attn.py ) 2>&1 | grep -v amdgpu | tee $O/attn_isolated.txt
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "attention or attn" 2>&1 | tail -3 | tee $O/pytest_attn.txt
