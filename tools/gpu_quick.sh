#!/bin/bash
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "attention or layernorm" 2>&1 | tail -2
timeout 300 python tools/bench_attn.py 2>&1 | grep -v amdgpu
