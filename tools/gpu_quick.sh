#!/bin/bash
# scratch: ABLATION (wrong results): K-strided fragments read with plain ds_read_b128
timeout 600 python tools/bench_gemm.py 2>&1 | grep "wgrad\|NN" | grep -v "s=1)"
for args in "3072 768 15104 TN 6 128 3"; do timeout 120 python tools/gemm_timeline.py $args 2>&1 | grep -v amdgpu.ids | head -6; done
