#!/bin/bash
mkdir -p gpurun_out/r3p
rm -f gpurun_out/ab/ab.txt
bash tools/gpu_ab.sh "MMTG_LMHEAD_GROUP=0" "MMTG_LMHEAD_GROUP=1" "MMTG_LMHEAD_GROUP_SPLITS=2"
cp gpurun_out/ab/ab.txt gpurun_out/r3p/ab_lmhead_group.txt
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_ddp_gpu.py -x -q --no-header -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/r3p/pytest_model.txt
python tools/determinism_probe.py 2>&1 | tail -6 | tee gpurun_out/r3p/determinism.txt
