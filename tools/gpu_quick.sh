#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "greedy or inference" > gpurun_out/model2.log 2>&1; echo "exit $?" >> gpurun_out/model2.log; tail -5 gpurun_out/model2.log
timeout 600 python tools/bench_gemm.py > gpurun_out/gemm_shapes.log 2>&1; tail -40 gpurun_out/gemm_shapes.log
