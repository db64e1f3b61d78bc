#!/bin/bash
mkdir -p gpurun_out/small
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -k "adamw or optim or loss or fused_train or full_size_training" -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -8 | tee gpurun_out/small/pytest.txt
python tools/step_breakdown.py 5 2>&1 | grep -E "instrumented|adamw|loss_fwd|loss_bwd|sumsq" | tee gpurun_out/small/breakdown.txt
