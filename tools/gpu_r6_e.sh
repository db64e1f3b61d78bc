#!/bin/bash
# Round 6, call E: the hybrid mode (bf16x3 forward + bf16 backward) -- parity tests, speed; HIP_FORCE_DEV_KERNARG set from inside Python.
mkdir -p gpurun_out/r6e
E=gpurun_out/r6e
timeout 1500 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "bf16x3f" 2>&1 | tail -15 | tee $E/pytest_bf16x3f.txt
for m in bf16x3f bf16x3 bf16; do timeout 300 python tools/bench_x3.py $m 64 10 2>&1 | grep -v amdgpu.ids | tee -a $E/bench_modes.txt; done
for ka in 0 default 0 default; do
  if [ $ka = 0 ]; then export HIP_FORCE_DEV_KERNARG=0; else unset HIP_FORCE_DEV_KERNARG; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --primary-only --no-roofline 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    print('train shell HIP_FORCE_DEV_KERNARG=$ka', d['value'], 'tok/s', d['ms_per_step'], 'ms/step')
" | tee -a $E/train_dev_kernarg_from_python.txt
done
unset HIP_FORCE_DEV_KERNARG
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_x3_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "bf16x3 and not bf16x3f and (forward or fused_train or reproducible)" 2>&1 | tail -6 | tee $E/pytest_x3_regress.txt
