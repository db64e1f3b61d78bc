#!/bin/bash
O=gpurun_out/p23; mkdir -p $O
timeout 1200 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "gemm or eight_phase or wgrad or layout" 2>&1 | tail -3 | tee $O/pytest_gemm.txt
mkdir -p gpurun_ab/base/tools; cp tools/bench_gemm.py gpurun_ab/base/tools/ 2>/dev/null
( echo "== base"; NTSET=1 python gpurun_ab/base/tools/bench_gemm.py; echo "== new"; NTSET=1 python tools/bench_gemm.py ) 2>&1 | grep -v amdgpu | tee $O/per_shape.txt
bash tools/gpu_ab.sh "" 2>&1 | tail -2 | tee $O/step_ab.txt
python gpurun_ab/base/bench.py --no-cpu-baseline --no-decode --no-check --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('BASE ms/step %.3f' % d['ms_per_step'], r['per_category_ms_per_step'])" | tee -a $O/step_ab.txt
bash tools/gpu_ab.sh "" 2>&1 | tail -2 | tee -a $O/step_ab.txt
