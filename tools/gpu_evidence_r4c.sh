#!/bin/bash
# Round 4 evidence, pass 3 (attention work of the second half of the round): isolated attention launches against the base
# build (gpurun_ab/base = the commit of the r04_v6 evidence set), the timelines of the three whole-head kernels, the attention
# tests under the A/B switches.
O=gpurun_out/ev4c; mkdir -p $O
mkdir -p gpurun_ab/base/tools; cp tools/bench_attn.py gpurun_ab/base/tools/ 2>/dev/null
( echo "== base (r04_v6 sources)"; python gpurun_ab/base/tools/bench_attn.py; echo "== final"; python tools/bench_attn.py
  echo "== base (r04_v6 sources)"; python gpurun_ab/base/tools/bench_attn.py; echo "== final"; python tools/bench_attn.py
  echo "== final, MMTG_ATTN_KV2=0 (16-key dK/dV builds)"; MMTG_ATTN_KV2=0 python tools/bench_attn.py
  echo "== final, MMTG_ATTN_BWD_FORK=1 (dQ kernel on a second stream)"; MMTG_ATTN_BWD_FORK=1 python tools/bench_attn.py ) 2>&1 | grep -v amdgpu | tee $O/attn_isolated_ab.txt
( python tools/attn_timeline.py 0.1; python tools/attn_bwd_timeline.py 0.1; echo "---- without dropout"; python tools/attn_bwd_timeline.py 0.0 ) 2>&1 | grep -v amdgpu | tee $O/attn_timelines.txt
( for v in "MMTG_ATTN_KV2=0" "MMTG_ATTN_BWD_FORK=1" "MMTG_ATTN_KV_NW=8"; do echo "== $v"; env $v timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "attention or attn or bit_reproducible or gradients" 2>&1 | tail -2; done ) | tee $O/attn_tests_under_switches.txt
bash tools/gpu_ab.sh "" 2>&1 | tail -2 | tee $O/step_ab.txt
python gpurun_ab/base/bench.py --no-cpu-baseline --no-decode --no-check --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('BASE (r04_v6 sources) ms/step %.3f' % d['ms_per_step'], r['per_category_ms_per_step'])" | tee -a $O/step_ab.txt
bash tools/gpu_ab.sh "" 2>&1 | tail -2 | tee -a $O/step_ab.txt
