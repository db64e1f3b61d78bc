#!/bin/bash
# Round 4 evidence, pass 1: the whole GPU suite + smoke, counter passes (GEMM family inside the bench; the FUSED decode step at
# --decode-len 64), rocprofv3 kernel statistics (train, decode, configs[4]), in-situ step breakdown, determinism probe.
# Copy the *_pmc_*.json files into profiles/ afterwards, then run tools/gpu_evidence_r4b.sh (the bench lines pick the traffic
# figures up by kernel-source sha).
mkdir -p gpurun_out/ev4
E=gpurun_out/ev4
export MMTG_TEST_REPORT=$(pwd)/$E/test_report.jsonl
rm -f $MMTG_TEST_REPORT
[ -n "$SKIP_TESTS" ] || timeout 2400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -4 | tee $E/pytest_gpu.txt
unset MMTG_TEST_REPORT
[ -n "$SKIP_TESTS" ] || timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $E/smoke.txt
timeout 900 bash tools/gpu_pmc_bench.sh > $E/pmc_bench.txt 2>&1; tail -1 $E/pmc_bench.txt
cp gpurun_out/bench_pmc_gemm_traffic.json $E/bench_pmc_gemm_traffic.json 2>/dev/null
timeout 900 bash tools/gpu_pmc_decode_r4.sh 64 > $E/pmc_decode_fused_len64.txt 2>&1; tail -1 $E/pmc_decode_fused_len64.txt | cut -c1-500
cp gpurun_out/decode_pmc_traffic_fused_len64.json $E/ 2>/dev/null
cat gpurun_out/pmc_decode_r4/rc.txt > $E/pmc_decode_rc.txt 2>/dev/null
timeout 900 bash tools/gpu_prof.sh > $E/prof.txt 2>&1
cp gpurun_out/prof/bench_kernel_stats.csv $E/rocprofv3_kernel_stats.csv 2>/dev/null
timeout 600 bash tools/gpu_decode_prof.sh > $E/decode_prof.txt 2>&1
cp $(find gpurun_out/dec/prof -name "*kernel_stats.csv" | head -1) $E/decode_rocprofv3_kernel_stats.csv 2>/dev/null
timeout 900 bash tools/gpu_medium_prof.sh > $E/medium_prof.txt 2>&1
cp $(find gpurun_out/med/prof -name "*kernel_stats.csv" | head -1) $E/medium_rocprofv3_kernel_stats.csv 2>/dev/null
python tools/step_breakdown.py 5 > $E/step_breakdown.txt 2>&1; head -3 $E/step_breakdown.txt
( FULL=1 TRIALS=12 timeout 900 python tools/determinism_probe.py ) 2>&1 | grep -v amdgpu | tail -14 > $E/determinism_full_size_12_trials.txt; tail -3 $E/determinism_full_size_12_trials.txt
