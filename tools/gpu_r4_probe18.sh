#!/bin/bash
O=gpurun_out/p18; mkdir -p $O
( python tools/attn_bwd_timeline.py 0.1; python tools/attn_bwd_timeline.py 0.0 ) 2>&1 | grep -v amdgpu | tee $O/attn_bwd_timeline.txt
python tools/bench_attn.py 2>&1 | grep -v amdgpu | tee $O/attn_isolated.txt
