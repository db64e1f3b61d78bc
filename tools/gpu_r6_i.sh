#!/bin/bash
mkdir -p gpurun_out/r6i
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "differentiates" 2>&1 | tail -40 | tee gpurun_out/r6i/alone.txt
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "bit_reproducible or differentiates or with_dropout_matches" 2>&1 | grep -E "^E  |passed|failed|assert" | head -40 | tee gpurun_out/r6i/with_neighbours.txt
