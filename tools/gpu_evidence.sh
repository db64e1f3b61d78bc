#!/bin/bash
# One call that refreshes every piece of evidence kept under profiles/ (copy the files afterwards):
# full -m gpu test run + smoke, rocprofv3 kernel stats of the bench, PMC HBM bytes per GEMM launch,
# in-situ step breakdown, per-shape GEMM tables (warm / HBM-cold), bench lines (train, decode).
mkdir -p gpurun_out/ev
timeout 1800 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/ev/pytest_gpu.txt
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/ev/smoke.txt
bash tools/gpu_pmc_bench.sh > gpurun_out/ev/pmc_bench.txt 2>&1; tail -1 gpurun_out/ev/pmc_bench.txt
cp gpurun_out/bench_pmc_gemm_traffic.json gpurun_out/ev/bench_pmc_gemm_traffic.json 2>/dev/null   # copy to profiles/r01_vN_bench_pmc_gemm_traffic.json afterwards
bash tools/gpu_prof.sh > gpurun_out/ev/prof.txt 2>&1
python tools/step_breakdown.py 5 > gpurun_out/ev/step_breakdown.txt 2>&1; head -3 gpurun_out/ev/step_breakdown.txt
python tools/bench_gemm.py > gpurun_out/ev/gemm_per_shape.txt 2>&1
echo "---- HBM-cold" >> gpurun_out/ev/gemm_per_shape.txt
COLD=1 python tools/bench_gemm.py >> gpurun_out/ev/gemm_per_shape.txt 2>&1
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/ev/bench_default.json; cut -c1-600 gpurun_out/ev/bench_default.json
timeout 900 python bench.py --mode decode 2>/dev/null | tail -1 > gpurun_out/ev/bench_decode.json; cut -c1-400 gpurun_out/ev/bench_decode.json
