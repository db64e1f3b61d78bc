#!/bin/bash
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "full_size" 2>&1 | tail -12
