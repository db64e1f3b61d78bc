#!/bin/bash
mkdir -p gpurun_out/r6l
E=gpurun_out/r6l
timeout 1500 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "bf16x3f" 2>&1 | tail -8 | tee $E/pytest_bf16x3f.txt
for e in 0 1 0 1; do MMTG_HYBRID_ENC_BF16=$e timeout 300 python tools/bench_x3.py bf16x3f 64 10 2>&1 | grep -v amdgpu.ids | head -3 | sed "s/^/ENC_BF16=$e /" | tee -a $E/enc_bf16_ab.txt; done
MODE=bf16x3f timeout 400 python tools/train_curve.py 3000 2>&1 | grep -v amdgpu.ids | tee $E/train_curve_3000_bf16x3f.txt | tail -3
