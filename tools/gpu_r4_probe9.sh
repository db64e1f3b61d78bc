#!/bin/bash
# round 4, probe 9: deterministic reductions -- op + model tests, determinism probe at full size, step A/B against the base tree
mkdir -p gpurun_out/p9
O=gpurun_out/p9
export MMTG_TEST_REPORT=$(pwd)/$O/test_report.jsonl
timeout 2400 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_ddp_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -8 > $O/pytest.txt; cat $O/pytest.txt
( FULL=1 TRIALS=6 NAMES=1 timeout 900 python tools/determinism_probe.py ) 2>&1 | grep -v amdgpu | tail -30 > $O/determinism_full.txt; cat $O/determinism_full.txt
( TRIALS=4 NAMES=1 STAGE=1 timeout 600 python tools/determinism_probe.py ) 2>&1 | grep -v amdgpu | tail -12 > $O/determinism_small.txt; cat $O/determinism_small.txt
bash tools/gpu_ab.sh "" "MMTG_LN_FINALIZE_ATOMIC=1" 2>&1 | tail -4 > $O/step_ab.txt; cat $O/step_ab.txt
python gpurun_ab/base/bench.py --no-cpu-baseline --no-decode --no-check --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('BASE (HEAD before this change) ms/step %.3f' % d['ms_per_step'], r['per_category_ms_per_step'])" | tee -a $O/step_ab.txt
python bench.py --no-cpu-baseline --no-decode --no-check --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('NEW ms/step %.3f' % d['ms_per_step'], r['per_category_ms_per_step'])" | tee -a $O/step_ab.txt
