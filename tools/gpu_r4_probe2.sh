#!/bin/bash
# round 4, probe 2: per-K-tile time of the eight-phase kernel vs row tile / N / column-block budget; new decoder parity tests
mkdir -p gpurun_out/p2
O=gpurun_out/p2
tl() { echo "=== $*"; env "$@" 2>&1 | grep -v "amdgpu.ids\|bin:\|alive\|distinct"; }
{
tl MMTG_GEMM_P8_ROWS=256 python tools/gemm_timeline.py 15104 3072 768 NT 0
tl MMTG_GEMM_P8_ROWS=192 python tools/gemm_timeline.py 15104 3072 768 NT 0
tl MMTG_GEMM_P8_ROWS=256 python tools/gemm_timeline.py 15104 2304 768 NT 0
tl MMTG_GEMM_P8_ROWS=256 python tools/gemm_timeline.py 15104 1536 768 NT 0
tl MMTG_GEMM_P8_ROWS=256 python tools/gemm_timeline.py 15104 768 768 NT 0
tl MMTG_GEMM_P8_ROWS=256 MMTG_GEMM_CB_KB=0 python tools/gemm_timeline.py 15104 3072 768 NT 0
tl MMTG_GEMM_P8_ROWS=256 MMTG_GEMM_CB_KB=1024 python tools/gemm_timeline.py 15104 3072 768 NT 0
tl MMTG_GEMM_P8_ROWS=256 MMTG_GEMM_CB_KB=4096 python tools/gemm_timeline.py 15104 3072 768 NT 0
tl MMTG_GEMM_P8_ROWS=256 python tools/gemm_timeline.py 8192 8192 8192 NT 0
tl MMTG_GEMM_P8_ROWS=256 python tools/gemm_timeline.py 4096 3072 768 NT 0
} > $O/timelines_rows_n_cb.txt 2>&1
export MMTG_TEST_REPORT=$(pwd)/$O/test_report.jsonl
timeout 1200 python -m pytest tests/test_decode_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "teacher or fused_decode_step or kv_cache" 2>&1 | tail -15 > $O/pytest_decoder.txt
cat $O/pytest_decoder.txt
