#!/bin/bash
# Round 6, call A: the one-launch decode MLP -- op tests, stand-alone timeline, token-step A/B -- plus the new host-side tests.
mkdir -p gpurun_out/r6b
E=gpurun_out/r6b
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "decode_mlp or decode_gemm_ln_fold" 2>&1 | tail -15 | tee $E/pytest_mlp_ops.txt
timeout 300 python tools/decode_mlp_timeline.py 256 2>&1 | tee $E/decode_mlp_timeline_m256.txt
timeout 200 python tools/decode_mlp_timeline.py 128 2>&1 | tail -30 > $E/decode_mlp_timeline_m128.txt
for cfg in "0 sc1" "1 sc1" "1 plain" "0 sc1" "1 plain"; do
  set -- $cfg
  MMTG_DECODE_MLP=$1 MMTG_DECODE_MLP_HANDOFF=$2 timeout 400 python bench.py --mode decode --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>$E/err_$1_$2.log | python -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    print('MLP=$1 handoff=$2', d['value'], 'tok/s', d['config']['us_per_token_step'], 'us/step', d['config']['once_per_generation_ms'], 'ms once', d['check'])
" | tee -a $E/decode_ab.txt
done
timeout 1500 python -m pytest tests/test_decode_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "teacher_forced or fused_decode_step or x3_engine or full_size_batched" 2>&1 | tail -15 | tee $E/pytest_decode.txt

