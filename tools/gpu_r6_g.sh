#!/bin/bash
mkdir -p gpurun_out/r6g
E=gpurun_out/r6g
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "differentiates_the_forward or (bit_reproducible and bf16x3f)" 2>&1 | tail -12 | tee $E/pytest_masks.txt
MODE=bf16x3f timeout 300 python tools/nan_hunt.py 400 2>&1 | grep -v amdgpu.ids | tee $E/nan_hunt_bf16x3f.txt | tail -25
MODE=bf16x3f PDROP=0 timeout 300 python tools/nan_hunt.py 400 2>&1 | grep -v amdgpu.ids | tee $E/nan_hunt_bf16x3f_nodrop.txt | tail -8
for es in 0 1 0 1; do
  MMTG_ENC_STREAMS=$es timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --primary-only --no-roofline 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    print('train MMTG_ENC_STREAMS=$es', d['value'], 'tok/s', d['ms_per_step'], 'ms/step')
" | tee -a $E/enc_streams_ab.txt
done
MMTG_ENC_STREAMS=1 timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "bit_reproducible or bf16_vs_oracle or fused_train_step or rnn_interlayer or encoder_sizes" 2>&1 | tail -6 | tee $E/pytest_enc_streams.txt
