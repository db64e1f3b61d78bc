#!/bin/bash
O=gpurun_out/p19; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "attention or attn" 2>&1 | tail -5 | tee $O/pytest_attn.txt
mkdir -p gpurun_ab/base/tools; cp tools/bench_attn.py gpurun_ab/base/tools/ 2>/dev/null
( echo "== base"; python gpurun_ab/base/tools/bench_attn.py; echo "== new"; python tools/bench_attn.py
  echo "== base"; python gpurun_ab/base/tools/bench_attn.py; echo "== new"; python tools/bench_attn.py ) 2>&1 | grep -v amdgpu | tee $O/attn_isolated.txt
( python tools/attn_bwd_timeline.py 0.1 ) 2>&1 | grep -v amdgpu | tee $O/attn_bwd_timeline.txt
