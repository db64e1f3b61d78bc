#!/bin/bash
mkdir -p gpurun_out/attn
export TMPDIR=/tmp
R=$(pwd)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/attn/prof -o attn -- python3 tools/bench_attn.py > gpurun_out/attn/prof.log 2>&1
f=$(find gpurun_out/attn/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 $f | cut -c1-200
find gpurun_out/attn/prof -name "*kernel_trace.csv" -delete
