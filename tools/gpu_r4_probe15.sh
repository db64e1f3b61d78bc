#!/bin/bash
# screening: the GPU suite under the round-4 switches
mkdir -p gpurun_out/p15
O=gpurun_out/p15
echo "== MMTG_DECODE_PERSIST=1: decode tests (persistent token step everywhere)" > $O/screen.txt
MMTG_DECODE_PERSIST=1 timeout 1500 python -m pytest tests/test_decode_gpu.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -4 >> $O/screen.txt
echo "== MMTG_DECODE_PERSIST=1 MMTG_DECODE_PERSIST_WGS=2" >> $O/screen.txt
MMTG_DECODE_PERSIST=1 MMTG_DECODE_PERSIST_WGS=2 timeout 1500 python -m pytest tests/test_decode_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "persistent or teacher or kv_cache or rules" 2>&1 | tail -4 >> $O/screen.txt
echo "== MMTG_GEMM_P8_ROWS=288: model + ops tests (every eight-phase product on 288-row tiles)" >> $O/screen.txt
MMTG_GEMM_P8_ROWS=288 timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -4 >> $O/screen.txt
echo "== MMTG_GEMM_P8T=1: weight gradients as K-strided eight-phase slabs (the de-waterfalled kernel)" >> $O/screen.txt
MMTG_GEMM_P8T=1 MMTG_WGRAD_GROUP=0 timeout 2400 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -4 >> $O/screen.txt
cat $O/screen.txt
