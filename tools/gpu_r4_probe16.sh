#!/bin/bash
# attention kernels: working tree vs the base build (gpurun_ab/base), isolated launches + the whole step + the attention tests
O=gpurun_out/p16; mkdir -p $O
mkdir -p gpurun_ab/base/tools; cp tools/bench_attn.py tools/bench_attn_rounds.py gpurun_ab/base/tools/ 2>/dev/null
( echo "== base"; python gpurun_ab/base/tools/bench_attn.py; echo "== new"; python tools/bench_attn.py
  echo "== base"; python gpurun_ab/base/tools/bench_attn.py; echo "== new"; python tools/bench_attn.py ) 2>&1 | grep -v amdgpu | tee $O/attn_isolated.txt
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "attention or attn" 2>&1 | tail -3 | tee $O/pytest_attn.txt
bash tools/gpu_ab.sh "" 2>&1 | tail -2 | tee $O/step_ab.txt
python gpurun_ab/base/bench.py --no-cpu-baseline --no-decode --no-check --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('BASE ms/step %.3f' % d['ms_per_step'], r['per_category_ms_per_step'])" | tee -a $O/step_ab.txt
bash tools/gpu_ab.sh "" 2>&1 | tail -1 | tee -a $O/step_ab.txt
