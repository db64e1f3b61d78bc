#!/bin/bash
O=gpurun_out/p20; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "attention or attn" 2>&1 | tail -5 | tee $O/pytest_attn.txt
( echo "== 16-key builds"; MMTG_ATTN_KV2=0 python tools/bench_attn.py; echo "== 32-key units"; python tools/bench_attn.py
  echo "== 16-key builds"; MMTG_ATTN_KV2=0 python tools/bench_attn.py; echo "== 32-key units"; python tools/bench_attn.py ) 2>&1 | grep -v amdgpu | tee $O/attn_isolated.txt
( python tools/attn_bwd_timeline.py 0.1 ) 2>&1 | grep -v amdgpu | head -14 | tee $O/attn_bwd_timeline.txt
