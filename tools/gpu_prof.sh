#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench command (program directly after --)
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$(pwd)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --primary-only --no-check > gpurun_out/prof_bench.log 2>&1
echo "rocprof exit $?" >> gpurun_out/prof_bench.log
tail -3 gpurun_out/prof_bench.log
find gpurun_out/prof -name "*stats*" | head; 
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f"
# drop the bulky per-dispatch trace, keep the summaries
find gpurun_out/prof -name "*kernel_trace.csv" -size +8M -delete
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.log 2>&1; tail -2 gpurun_out/bench_default.log
