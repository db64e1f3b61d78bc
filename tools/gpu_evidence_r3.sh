#!/bin/bash
# Round 3 evidence, pass 1 (counter passes + kernel statistics; copy the *_pmc_*.json files into profiles/ afterwards, then run
# tools/gpu_evidence_r3b.sh so that the bench lines pick the traffic figures up by kernel-source sha)
mkdir -p gpurun_out/ev3
export MMTG_TEST_REPORT=$(pwd)/gpurun_out/ev3/test_report.jsonl
rm -f $MMTG_TEST_REPORT
[ -n "$SKIP_TESTS" ] || timeout 2400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/ev3/pytest_gpu.txt
unset MMTG_TEST_REPORT
[ -n "$SKIP_TESTS" ] || timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/ev3/smoke.txt
timeout 900 bash tools/gpu_pmc_bench.sh > gpurun_out/ev3/pmc_bench.txt 2>&1; tail -1 gpurun_out/ev3/pmc_bench.txt
cp gpurun_out/bench_pmc_gemm_traffic.json gpurun_out/ev3/bench_pmc_gemm_traffic.json 2>/dev/null
timeout 900 bash tools/gpu_pmc_decode.sh > gpurun_out/ev3/pmc_decode.txt 2>&1; tail -1 gpurun_out/ev3/pmc_decode.txt | cut -c1-600
cp gpurun_out/decode_pmc_traffic.json gpurun_out/ev3/decode_pmc_traffic.json 2>/dev/null
timeout 900 bash tools/gpu_prof.sh > gpurun_out/ev3/prof.txt 2>&1
cp gpurun_out/prof/bench_kernel_stats.csv gpurun_out/ev3/rocprofv3_kernel_stats.csv 2>/dev/null
timeout 600 bash tools/gpu_decode_prof.sh > gpurun_out/ev3/decode_prof.txt 2>&1
cp $(find gpurun_out/dec/prof -name "*kernel_stats.csv" | head -1) gpurun_out/ev3/decode_rocprofv3_kernel_stats.csv 2>/dev/null
timeout 900 bash tools/gpu_medium_prof.sh > gpurun_out/ev3/medium_prof.txt 2>&1
cp $(find gpurun_out/med/prof -name "*kernel_stats.csv" | head -1) gpurun_out/ev3/medium_rocprofv3_kernel_stats.csv 2>/dev/null
python tools/step_breakdown.py 5 > gpurun_out/ev3/step_breakdown.txt 2>&1; head -3 gpurun_out/ev3/step_breakdown.txt
