#!/bin/bash
# round 5: rocprofv3 kernel stats of the bf16x3 decode step and the bf16x3 train step
mkdir -p gpurun_out/r5c
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5c/dec -o dec -- python3 $R/bench.py --mode decode --dtype bf16x3 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/r5c/dec.log 2>&1
cd $R
f=$(find gpurun_out/r5c/dec -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r5c/decode_x3_kernel_stats.csv && head -16 $f | cut -c1-200
find gpurun_out/r5c -name "*kernel_trace.csv" -delete
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5c/tr -o tr -- python3 $R/tools/bench_x3.py bf16x3 64 5 > $R/gpurun_out/r5c/train.log 2>&1
cd $R
f=$(find gpurun_out/r5c/tr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r5c/train_x3_kernel_stats.csv && head -30 $f | cut -c1-200
find gpurun_out/r5c -name "*kernel_trace.csv" -delete
tail -3 gpurun_out/r5c/train.log
