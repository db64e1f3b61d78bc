#!/bin/bash
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -3
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo
done
