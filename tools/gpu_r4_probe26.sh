#!/bin/bash
O=gpurun_out/p26; mkdir -p $O
timeout 600 env MMTG_WGRAD_TOUCH=2 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "wgrad" 2>&1 | tail -3 | tee $O/pytest_wgrad_touch.txt
( for t in 0 1 2 3 4 6; do echo "== MMTG_WGRAD_TOUCH=$t"; MMTG_WGRAD_TOUCH=$t python tools/bench_wgrad_group.py 2>&1 | grep -v amdgpu | tail -4; done ) | tee $O/wgrad_touch_isolated.txt
bash tools/gpu_ab.sh "" "MMTG_WGRAD_TOUCH=2" "MMTG_WGRAD_TOUCH=3" 2>&1 | tail -6 | tee $O/step_ab.txt
