#!/bin/bash
mkdir -p gpurun_out/r3s
timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q --no-header -p no:cacheprovider -k "attention" 2>&1 | tail -12 | tee gpurun_out/r3s/pytest_attn.txt
