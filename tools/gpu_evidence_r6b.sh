#!/bin/bash
# Round 6 evidence, pass 2 (after the PMC files of pass 1 were copied into profiles/): the default bench line (all objects), the decode
# lines of both modes, the 3000-step memorisation curves of the three fast modes, the full-size determinism probe, the two-rank
# rehearsal on one GPU (gloo through the host: the N > 1 code path incl. the per-bucket timeline).
mkdir -p gpurun_out/ev6b
E=gpurun_out/ev6b
timeout 1500 python bench.py > $E/bench_default.json 2> $E/bench_default.err; tail -c 400 $E/bench_default.json
timeout 600 python bench.py --mode decode > $E/bench_decode_bf16.json 2>> $E/bench_default.err
timeout 600 python bench.py --mode decode --dtype bf16x3 > $E/bench_decode_bf16x3.json 2>> $E/bench_default.err
for m in bf16x3f bf16x3 bf16; do MODE=$m timeout 400 python tools/train_curve.py 3000 2>&1 | grep -v amdgpu.ids > $E/train_curve_3000_$m.txt; tail -2 $E/train_curve_3000_$m.txt; done
timeout 600 python tools/determinism_probe.py 2>&1 | grep -v amdgpu.ids | tail -6 | tee $E/determinism_probe.txt
MMTG_BENCH_ONE_GPU_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --primary-only > $E/bench_2rank_rehearsal.json 2> $E/bench_2rank.err; tail -c 1200 $E/bench_2rank_rehearsal.json
