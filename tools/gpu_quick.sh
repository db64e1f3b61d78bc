#!/bin/bash
timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "decode or gemm" 2>&1 | tail -4
MMTG_DECODE_PROF=1 timeout 900 python bench.py --mode decode --steps 1 --warmup 1 2>&1 | tail -1
timeout 900 python bench.py --mode decode --steps 3 --warmup 1 2>&1 | tail -1 | cut -c1-300
timeout 600 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | grep -v "s=1)" | tail -4
