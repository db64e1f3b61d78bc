#!/bin/bash
# Round 6 evidence, pass 1 (kernel sources final): the whole GPU suite + smoke; counter passes (GEMM family inside the bf16 bench; the
# decode token step at the benchmarked 128 positions, bf16 and bf16x3; MFMA / LDS utilisation of the bf16 step's kernels); rocprofv3 kernel
# statistics (bf16 / bf16x3f / bf16x3 train, bf16 / bf16x3 decode).  Copy the *_pmc_*.json files into profiles/ afterwards, then run
# tools/gpu_evidence_r6b.sh (the bench lines pick the traffic figures up by kernel-source sha).
mkdir -p gpurun_out/ev6
E=gpurun_out/ev6
export MMTG_TEST_REPORT=$(pwd)/$E/test_report.jsonl
rm -f $MMTG_TEST_REPORT
[ -n "$SKIP_TESTS" ] || ( time timeout 3300 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider ) 2>&1 | tail -8 | tee $E/pytest_gpu.txt
unset MMTG_TEST_REPORT
[ -n "$SKIP_TESTS" ] || timeout 600 python __graft_entry__.py smoke 2>&1 | tail -4 | tee $E/smoke.txt
timeout 900 bash tools/gpu_pmc_bench.sh > $E/pmc_bench.txt 2>&1; tail -1 $E/pmc_bench.txt | cut -c1-600
cp gpurun_out/bench_pmc_gemm_traffic.json $E/bench_pmc_gemm_traffic.json 2>/dev/null
for dt in bf16 bf16x3; do
  timeout 1300 bash tools/gpu_pmc_decode_r5.sh 128 $dt > $E/pmc_decode_${dt}_len128.txt 2>&1; tail -1 $E/pmc_decode_${dt}_len128.txt | cut -c1-700
  cp gpurun_out/decode_pmc_traffic_${dt}_len128.json $E/ 2>/dev/null
  cp gpurun_out/pmc_decode_r5_$dt/rc.txt $E/pmc_decode_${dt}_attempts_rc.txt 2>/dev/null
done
timeout 1500 bash tools/gpu_pmc_util_r6.sh > $E/pmc_mfma_lds_utilisation_bf16.txt 2>&1; tail -22 $E/pmc_mfma_lds_utilisation_bf16.txt
export TMPDIR=/tmp
R=$(pwd)
prof() {   # name, program args...
  local name=$1; shift
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$E/prof_$name -o p -- python3 "$@" > $R/$E/prof_$name.log 2>&1 )
  local f=$(find $E/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $E/${name}_rocprofv3_kernel_stats.csv && head -6 $f | cut -c1-160
  rm -rf $E/prof_$name
}
prof bf16_train $R/bench.py --steps 10 --warmup 4 --no-cpu-baseline --primary-only --no-check
# (the same program with every launch alone on the GPU: the schedule the bench line's instrumented roofline pass uses -- the product
#  schedule above runs 8 grouped weight-gradient launches beside the backward's tail, where their durations include the sharing)
export MMTG_WGRAD_TAIL=0
prof bf16_train_unshared $R/bench.py --steps 10 --warmup 4 --no-cpu-baseline --primary-only --no-check
unset MMTG_WGRAD_TAIL
prof bf16x3f_train $R/tools/bench_x3.py bf16x3f 64 10
prof bf16x3_train $R/tools/bench_x3.py bf16x3 64 10
prof bf16_decode $R/bench.py --mode decode --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
prof bf16x3_decode $R/bench.py --mode decode --dtype bf16x3 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
