#!/bin/bash
mkdir -p gpurun_out/r6j
timeout 600 python -m pytest tests/test_x3_gpu.py tests/test_ddp_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "element_dropout_mask or bf16_gradient_exchange" 2>&1 | tail -25 | tee gpurun_out/r6j/pytest_new.txt
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -6 | tee gpurun_out/r6j/smoke.txt
