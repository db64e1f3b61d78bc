#!/bin/bash
mkdir -p gpurun_out/attn
timeout 900 python -m pytest tests/test_ops_gpu.py -k "attention and not alpha and not decode" -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -30 | tee gpurun_out/attn/pytest.txt
python tools/bench_attn.py 2>&1 | tee gpurun_out/attn/bench_new.txt
python tools/attn_timeline.py 0.1 2>&1 | tee gpurun_out/attn/timeline.txt
bash tools/gpu_attn_prof.sh
