#!/bin/bash
timeout 900 python -m pytest tests/test_decode_gpu.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -3
MMTG_DECODE_PROF=1 timeout 900 python bench.py --mode decode --steps 1 --warmup 1 2>&1 | tail -1
