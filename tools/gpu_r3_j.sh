#!/bin/bash
mkdir -p gpurun_out/r3j
timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_ops_gpu.py -x -q --no-header -p no:cacheprovider -k "decode or logits_process or kv_cache or generate or sample" 2>&1 | tail -4 | tee gpurun_out/r3j/pytest_decode.txt
timeout 600 python -m pytest tests/test_model_gpu.py -x -q --no-header -p no:cacheprovider -k "greedy or sample_sequence or infer" 2>&1 | tail -3 | tee gpurun_out/r3j/pytest_model_decode.txt
timeout 600 python bench.py --mode decode --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3j/bench_decode.json; cut -c1-900 gpurun_out/r3j/bench_decode.json
