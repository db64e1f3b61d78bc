#!/bin/bash
mkdir -p gpurun_out/p13
O=gpurun_out/p13
( echo "== NTSET (no waterfall loops)"; NTSET=1 python tools/bench_gemm.py
  echo "== TNSET eight-phase K-strided slabs"; MMTG_GEMM_P8T=1 SLAB=2 TNSET=1 python tools/bench_gemm.py 2>&1 | grep wgrad ) 2>&1 | grep -v amdgpu > $O/per_shape.txt; cat $O/per_shape.txt
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "eight_phase" 2>&1 | tail -3
bash tools/gpu_ab.sh "" "MMTG_GEMM_P8T=1" 2>&1 | tail -4 > $O/step_ab.txt; cat $O/step_ab.txt
python gpurun_ab/base/bench.py --no-cpu-baseline --no-decode --no-check --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('BASE ms/step %.3f' % d['ms_per_step'], r['per_category_ms_per_step'])" | tee -a $O/step_ab.txt
MMTG_EXTRA_DEFS=-DMMTG_P8_PHASE_TRACE python -m mmtg_amd.build --force --jobs 16 2>&1 | tail -1
timeout 300 python tools/p8_phase_trace.py 2>&1 | grep -v amdgpu > $O/p8_phase_trace_after.txt; cat $O/p8_phase_trace_after.txt
