#!/bin/bash
O=gpurun_out/p27; mkdir -p $O
( for tk in 15104 7552 3776; do echo "== TOKENS=$tk"; TOKENS=$tk SPLITS=2 SPLITS8=2 python tools/bench_wgrad_group.py 2>&1 | grep -v amdgpu | grep -v "slab launches"; done ) | tee $O/wgrad_mall_residency.txt
