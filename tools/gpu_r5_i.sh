#!/bin/bash
mkdir -p gpurun_out/r5i
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5i/tr -o tr -- python3 $R/tools/bench_x3.py bf16x3 64 5 > $R/gpurun_out/r5i/train.log 2>&1
cd $R
f=$(find gpurun_out/r5i/tr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r5i/train_x3_kernel_stats.csv && head -32 $f | cut -c1-150
find gpurun_out/r5i -name "*kernel_trace.csv" -delete
