#!/bin/bash
# refresh the PMC traffic measurement for the current kernel sources + run the GEMM / DDP tests
mkdir -p gpurun_out/ev
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ddp_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "gemm or ddp" 2>&1 | tail -3 | tee gpurun_out/ev/pytest_gemm_ddp.txt
bash tools/gpu_pmc_bench.sh > gpurun_out/ev/pmc_bench.txt 2>&1; tail -1 gpurun_out/ev/pmc_bench.txt
cp gpurun_out/bench_pmc_gemm_traffic.json gpurun_out/ev/bench_pmc_gemm_traffic.json
