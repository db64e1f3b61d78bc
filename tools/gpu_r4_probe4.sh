#!/bin/bash
mkdir -p gpurun_out/p4
O=gpurun_out/p4
timeout 900 python tools/decode_lanes_threads.py --lanes 1,2,4,8 2>&1 | grep -v amdgpu > $O/decode_lanes_threads.txt; cat $O/decode_lanes_threads.txt
export MMTG_TEST_REPORT=$(pwd)/$O/test_report.jsonl
timeout 1500 python -m pytest tests/test_decode_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -5 > $O/pytest_decode.txt; cat $O/pytest_decode.txt
tl() { echo "=== $*"; env "$@" 2>&1 | grep -v "amdgpu.ids\|bin:\|alive\|distinct"; }
{
tl MMTG_GEMM_P8_ROWS=256 python tools/gemm_timeline.py 15104 3072 768 NT 0
tl MMTG_GEMM_P8_ROWS=192 python tools/gemm_timeline.py 15104 768 3072 NT 0
tl MMTG_GEMM_P8_ROWS=256 python tools/gemm_timeline.py 15104 768 768 NT 0
tl MMTG_GEMM_P8_ROWS=256 python tools/gemm_timeline.py 8192 8192 8192 NT 0
} > $O/timelines_clock.txt 2>&1; cat $O/timelines_clock.txt
