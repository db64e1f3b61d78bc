#!/bin/bash
mkdir -p gpurun_out/dec
export TMPDIR=/tmp
R=$(pwd)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dec/prof -o dec -- python3 bench.py --mode decode --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/dec/prof.log 2>&1
f=$(find gpurun_out/dec/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -24 $f | cut -c1-220
find gpurun_out/dec/prof -name "*kernel_trace.csv" -delete
