#!/bin/bash
# One call that refreshes every piece of evidence kept under profiles/ (copy the files afterwards):
# full -m gpu test run + smoke, rocprofv3 kernel stats of the bench, PMC HBM bytes per GEMM launch,
# in-situ step breakdown, per-shape GEMM tables (warm / HBM-cold), bench lines (train bf16 + decode object, train f32,
# decode mode), attention timeline, rocprofv3 stats of the decode bench.
mkdir -p gpurun_out/ev
export MMTG_TEST_REPORT=$(pwd)/gpurun_out/ev/test_report.jsonl
rm -f $MMTG_TEST_REPORT
timeout 2400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/ev/pytest_gpu.txt
unset MMTG_TEST_REPORT
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/ev/smoke.txt
bash tools/gpu_pmc_bench.sh > gpurun_out/ev/pmc_bench.txt 2>&1; tail -1 gpurun_out/ev/pmc_bench.txt
cp gpurun_out/bench_pmc_gemm_traffic.json gpurun_out/ev/bench_pmc_gemm_traffic.json 2>/dev/null   # copy to profiles/rNN_vN_bench_pmc_gemm_traffic.json afterwards
bash tools/gpu_prof.sh > gpurun_out/ev/prof.txt 2>&1
cp gpurun_out/prof/bench_kernel_stats.csv gpurun_out/ev/rocprofv3_kernel_stats.csv 2>/dev/null
cp gpurun_out/prof/bench_domain_stats.csv gpurun_out/ev/rocprofv3_domain_stats.csv 2>/dev/null
python tools/step_breakdown.py 5 > gpurun_out/ev/step_breakdown.txt 2>&1; head -3 gpurun_out/ev/step_breakdown.txt
python tools/bench_gemm.py > gpurun_out/ev/gemm_per_shape.txt 2>&1
echo "---- the K-contiguous products as the trainer launches them (NT) + calibration squares" >> gpurun_out/ev/gemm_per_shape.txt
NTSET=1 python tools/bench_gemm.py >> gpurun_out/ev/gemm_per_shape.txt 2>&1
echo "---- HBM-cold" >> gpurun_out/ev/gemm_per_shape.txt
COLD=1 python tools/bench_gemm.py >> gpurun_out/ev/gemm_per_shape.txt 2>&1
COLD=1 NTSET=1 python tools/bench_gemm.py >> gpurun_out/ev/gemm_per_shape.txt 2>&1
{ python tools/gemm_timeline.py 15104 3072 768 NT 1; python tools/gemm_timeline.py 15104 768 3072 NT 3; python tools/gemm_timeline.py 15104 13440 768 NT 0; } > gpurun_out/ev/gemm_eight_phase_timeline.txt 2>&1
python tools/attn_timeline.py 0.1 > gpurun_out/ev/attn_fwd_timeline.txt 2>&1
python tools/bench_attn.py > gpurun_out/ev/attn_isolated.txt 2>&1
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/ev/bench_default.json; cut -c1-600 gpurun_out/ev/bench_default.json
timeout 900 python bench.py --dtype f32 --steps 5 --warmup 2 --no-cpu-baseline --no-decode 2>/dev/null | tail -1 > gpurun_out/ev/bench_f32.json; cut -c1-400 gpurun_out/ev/bench_f32.json
timeout 900 python bench.py --mode decode 2>/dev/null | tail -1 > gpurun_out/ev/bench_decode.json; cut -c1-400 gpurun_out/ev/bench_decode.json
bash tools/gpu_decode_prof.sh > gpurun_out/ev/decode_prof.txt 2>&1
cp gpurun_out/dec/prof/dec_kernel_stats.csv gpurun_out/ev/decode_rocprofv3_kernel_stats.csv 2>/dev/null
# configs[4] at its stated size on one GPU: bench line (eight-phase kernel on / off) + rocprofv3 stats + per-shape table
timeout 900 python bench.py --config medium --steps 8 --warmup 3 --no-decode 2>/dev/null | tail -1 > gpurun_out/ev/bench_medium.json; cut -c1-400 gpurun_out/ev/bench_medium.json
MMTG_GEMM_P8=0 timeout 900 python bench.py --config medium --steps 8 --warmup 3 --no-decode --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ev/bench_medium_no_eight_phase.json
MMTG_GEMM_P8=0 timeout 900 python bench.py --no-decode --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ev/bench_default_no_eight_phase.json
bash tools/gpu_medium_prof.sh > gpurun_out/ev/medium_prof.txt 2>&1
cp $(find gpurun_out/med/prof -name "*kernel_stats.csv" | head -1) gpurun_out/ev/medium_rocprofv3_kernel_stats.csv 2>/dev/null
cp gpurun_out/med/gemm_per_shape_medium.txt gpurun_out/ev/gemm_per_shape_medium.txt 2>/dev/null
