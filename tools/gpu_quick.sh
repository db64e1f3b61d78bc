#!/bin/bash
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "gemm" 2>&1 | tail -3
timeout 600 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | grep -v "s=1)"
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -3
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1700
