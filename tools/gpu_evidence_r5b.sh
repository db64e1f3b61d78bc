#!/bin/bash
# Round 5 evidence, pass 2 (after the PMC files of pass 1 were copied into profiles/): the default bench line (all objects), the decode
# lines of both modes, the two-rank rehearsal on one GPU (gloo through the host: the N > 1 code path incl. the CU-reservation tuning).
mkdir -p gpurun_out/ev5b
E=gpurun_out/ev5b
timeout 1500 python bench.py > $E/bench_default.json 2> $E/bench_default.err; tail -c 600 $E/bench_default.json
timeout 600 python bench.py --mode decode > $E/bench_decode_bf16.json 2>> $E/bench_default.err
timeout 600 python bench.py --mode decode --dtype bf16x3 > $E/bench_decode_bf16x3.json 2>> $E/bench_default.err
MMTG_BENCH_ONE_GPU_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --primary-only > $E/bench_2rank_rehearsal.json 2> $E/bench_2rank.err; tail -c 1500 $E/bench_2rank_rehearsal.json
