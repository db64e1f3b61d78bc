#!/bin/bash
O=gpurun_out/p25; mkdir -p $O
( echo "== one stream"; python tools/bench_attn.py; echo "== fork"; MMTG_ATTN_BWD_FORK=1 python tools/bench_attn.py ) 2>&1 | grep -v amdgpu | tee $O/attn_isolated.txt
bash tools/gpu_ab.sh "" "MMTG_ATTN_BWD_FORK=1" 2>&1 | tail -4 | tee $O/step_ab.txt
