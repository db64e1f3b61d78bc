#!/bin/bash
# round 4, probe 3: 288-row tiles (tests + timelines + per-shape), fused-decode counter traffic at 64 (and a try at 128), CU contention table, bench line with the medium object
mkdir -p gpurun_out/p3
O=gpurun_out/p3
export MMTG_TEST_REPORT=$(pwd)/$O/test_report.jsonl
timeout 1500 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "eight_phase or wgrad_group or gemm_layouts" 2>&1 | tail -8 > $O/pytest_gemm.txt
cat $O/pytest_gemm.txt
timeout 600 python -m pytest tests/test_decode_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "host_ratings or edge_cases" 2>&1 | tail -5 > $O/pytest_trainer.txt
cat $O/pytest_trainer.txt
tl() { echo "=== $*"; env "$@" 2>&1 | grep -v "amdgpu.ids\|bin:\|alive\|distinct"; }
{
tl MMTG_GEMM_P8_ROWS=288 python tools/gemm_timeline.py 15104 2304 768 NT 0
tl MMTG_GEMM_P8_ROWS=192 python tools/gemm_timeline.py 15104 2304 768 NT 0
tl MMTG_GEMM_P8_ROWS=288 python tools/gemm_timeline.py 15104 13440 768 NT 0
tl MMTG_GEMM_P8_ROWS=256 python tools/gemm_timeline.py 15104 13440 768 NT 0
tl MMTG_GEMM_P8_ROWS=288 python tools/gemm_timeline.py 15104 3072 768 NT 1
} > $O/timelines_288.txt 2>&1
( echo "--- default rule"; NTSET=1 python tools/bench_gemm.py; echo "--- MMTG_GEMM_P8_288=0"; MMTG_GEMM_P8_288=0 NTSET=1 python tools/bench_gemm.py ) 2>&1 | grep -v amdgpu > $O/ntset_288_ab.txt
python tools/ddp_contention.py --steps 10 > $O/ddp_contention.txt 2>&1
tail -5 $O/ddp_contention.txt
bash tools/gpu_pmc_decode_r4.sh 64 > $O/pmc_decode_64.log 2>&1; tail -3 $O/pmc_decode_64.log
cp gpurun_out/decode_pmc_traffic_fused_len64.json $O/ 2>/dev/null
bash tools/gpu_ab.sh "" "MMTG_GEMM_P8_288=0" 2>&1 | tail -4 > $O/step_ab_288.txt; cat $O/step_ab_288.txt
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
