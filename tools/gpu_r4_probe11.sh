#!/bin/bash
mkdir -p gpurun_out/p11
O=gpurun_out/p11
MMTG_FORCE_DDP=1 timeout 600 python bench.py --no-cpu-baseline --no-decode --steps 10 --warmup 3 > $O/bench_forced_ddp_world1.json 2> $O/bench_forced_ddp.err; tail -c 900 $O/bench_forced_ddp_world1.json; tail -3 $O/bench_forced_ddp.err
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ddp_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "beta or fuse or ddp or colsum or embed or adamw or optim" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "reproducible or forward_f32 or fused_train_step" 2>&1 | tail -3
python tools/step_breakdown.py 5 2>/dev/null | grep "beta_fuse_bwd\|instrumented"
