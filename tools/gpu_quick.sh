#!/bin/bash
# scratch: fused bias gradients (dGELU epilogue, attention backward)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "gemm_epilogues or attention" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1400
