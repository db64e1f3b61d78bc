#!/bin/bash
mkdir -p gpurun_out/attn
timeout 900 python -m pytest tests/test_ops_gpu.py -k "attention and not alpha and not decode" -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/attn/pytest.txt
MMTG_ATTN_KV4=1 timeout 900 python -m pytest tests/test_ops_gpu.py -k "attention and not alpha and not decode" -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -5 | tee -a gpurun_out/attn/pytest.txt
python tools/bench_attn.py 2>&1 | tee gpurun_out/attn/bench_new.txt
MMTG_ATTN_KV4=1 python tools/bench_attn.py 2>&1 | tee gpurun_out/attn/bench_kv4.txt
