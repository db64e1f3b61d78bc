#!/bin/bash
O=gpurun_out/p21; mkdir -p $O
bash tools/gpu_ab.sh "" 2>&1 | tail -2 | tee $O/step_ab.txt
python gpurun_ab/base/bench.py --no-cpu-baseline --no-decode --no-check --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('BASE ms/step %.3f' % d['ms_per_step'], r['per_category_ms_per_step'])" | tee -a $O/step_ab.txt
bash tools/gpu_ab.sh "" "MMTG_ATTN_KV2=0" 2>&1 | tail -4 | tee -a $O/step_ab.txt
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -5 | tee $O/pytest_gpu.txt
python tools/determinism_probe.py 2>&1 | grep -v amdgpu | tail -5 | tee $O/determinism.txt
