#!/bin/bash
mkdir -p gpurun_out/r6d
E=gpurun_out/r6d
timeout 200 python tools/decode_mlp_timeline.py 256 hot 2>&1 | grep -v amdgpu.ids | head -20 | tee $E/mlp_timeline_hot.txt
HIP_FORCE_DEV_KERNARG=1 timeout 60 ./tools/micro/node_floor 2>&1 | head -8 | tee $E/node_floor_dev_kernarg.txt
for ka in 0 1 0 1; do
  HIP_FORCE_DEV_KERNARG=$ka timeout 400 python bench.py --mode decode --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    print('HIP_FORCE_DEV_KERNARG=$ka', d['value'], 'tok/s', d['config']['us_per_token_step'], 'us/step', d['config']['once_per_generation_ms'], 'ms once')
" | tee -a $E/decode_dev_kernarg_ab.txt
done
for ka in 0 1 0 1; do
  HIP_FORCE_DEV_KERNARG=$ka timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --primary-only --no-roofline 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    print('train HIP_FORCE_DEV_KERNARG=$ka', d['value'], 'tok/s', d['ms_per_step'], 'ms/step')
" | tee -a $E/train_dev_kernarg_ab.txt
done
