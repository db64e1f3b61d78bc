#!/bin/bash
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "short_sequence" 2>&1 | tail -30
