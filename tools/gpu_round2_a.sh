#!/bin/bash
# round-2 first validation: all -m gpu tests (with the measured-bounds report), smoke, default bench line,
# forced-DDP bench (RCCL at world 1) with a rocprofv3 kernel trace
mkdir -p gpurun_out/r2a
export MMTG_TEST_REPORT=$(pwd)/gpurun_out/r2a/test_report.jsonl
rm -f $MMTG_TEST_REPORT
timeout 2400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -40 | tee gpurun_out/r2a/pytest_gpu.txt
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/r2a/smoke.txt
timeout 900 python bench.py 2> gpurun_out/r2a/bench_default.err | tail -1 > gpurun_out/r2a/bench_default.json; cut -c1-3000 gpurun_out/r2a/bench_default.json; tail -5 gpurun_out/r2a/bench_default.err
MMTG_FORCE_DDP=1 timeout 600 python bench.py --no-cpu-baseline --no-decode 2> gpurun_out/r2a/bench_ddp1.err | tail -1 > gpurun_out/r2a/bench_forced_ddp.json; cut -c1-600 gpurun_out/r2a/bench_forced_ddp.json; tail -3 gpurun_out/r2a/bench_ddp1.err
# rocprofv3 kernel trace of the forced-DDP bench: RCCL kernels vs the backward's kernels
export TMPDIR=/tmp
R=$(pwd)
export MMTG_FORCE_DDP=1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2a/ddp_prof -o ddp -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-decode --no-check > gpurun_out/r2a/ddp_prof.log 2>&1
unset MMTG_FORCE_DDP
f=$(find gpurun_out/r2a/ddp_prof -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 tools/ddp_overlap.py $f | tee gpurun_out/r2a/ddp_overlap.json
f2=$(find gpurun_out/r2a/ddp_prof -name "*kernel_stats.csv" | head -1); [ -n "$f2" ] && head -12 $f2
find gpurun_out/r2a/ddp_prof -name "*kernel_trace.csv" -size +8M -delete
