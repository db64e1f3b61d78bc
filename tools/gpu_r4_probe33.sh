#!/bin/bash
O=gpurun_out/p33; mkdir -p $O
MMTG_ATTN_Q2=1 timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "attention or attn" 2>&1 | tail -3 | tee $O/pytest_attn_q2.txt
( echo "== dQ kernel: 16-query tiles, 8 waves (default)"; python tools/bench_attn.py; echo "== MMTG_ATTN_Q2=1: 32-query units, 4 waves"; MMTG_ATTN_Q2=1 python tools/bench_attn.py
  echo "== default"; python tools/bench_attn.py; echo "== MMTG_ATTN_Q2=1"; MMTG_ATTN_Q2=1 python tools/bench_attn.py ) 2>&1 | grep -v amdgpu | tee $O/attn_q2_ab.txt
MMTG_ATTN_Q2=1 python tools/attn_bwd_timeline.py 0.1 2>&1 | grep -v amdgpu | tail -14 | tee $O/timeline_q2.txt
bash tools/gpu_ab.sh "" "MMTG_ATTN_Q2=1" 2>&1 | tail -4 | tee $O/step_ab.txt
