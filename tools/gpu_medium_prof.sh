#!/bin/bash
# configs[4] evidence: rocprofv3 kernel stats of the medium bench + per-shape GEMM table at GPT-2-medium widths
mkdir -p gpurun_out/med
export TMPDIR=/tmp
R=$(pwd)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/med/prof -o med -- python3 bench.py --config medium --steps 4 --warmup 2 --no-check > gpurun_out/med/prof.log 2>&1
f=$(find gpurun_out/med/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 $f | cut -c1-200
find gpurun_out/med/prof -name "*kernel_trace.csv" -delete
DMODEL=1024 TOKENS=16384 python tools/bench_gemm.py > gpurun_out/med/gemm_per_shape_medium.txt 2>&1; tail -30 gpurun_out/med/gemm_per_shape_medium.txt
