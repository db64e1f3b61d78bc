#!/bin/bash
mkdir -p gpurun_out/gather
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -k "bf16 or golden or gather or clipped or medium or additive or full_size_training or workspace" -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -12 | tee gpurun_out/gather/pytest.txt
bash tools/gpu_ab.sh "X=1" "MMTG_NO_GATHER=1"
python tools/step_breakdown.py 5 2>&1 | grep -E "instrumented|gather|embed_condition|N=512 K=2048|M=512 N=2048" | tee gpurun_out/gather/breakdown.txt
