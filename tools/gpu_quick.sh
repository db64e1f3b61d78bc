#!/bin/bash
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -3
for v in 0 1 0 1; do echo "== side stream $v"; MMTG_SIDE_STREAM=$v timeout 900 python bench.py --no-cpu-baseline --no-roofline 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*'; done
