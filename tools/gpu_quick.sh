#!/bin/bash
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "loss or model or train" 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"loss": [0-9.]*'
