#!/bin/bash
# Round 3 evidence, pass 2: the bench lines (profiles/ already holds this source sha's PMC traffic files)
mkdir -p gpurun_out/ev3
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/ev3/bench_default.json; cut -c1-500 gpurun_out/ev3/bench_default.json
timeout 900 python bench.py --mode decode 2>/dev/null | tail -1 > gpurun_out/ev3/bench_decode.json; cut -c1-300 gpurun_out/ev3/bench_decode.json
timeout 900 python bench.py --config medium --steps 8 --warmup 3 --no-decode 2>/dev/null | tail -1 > gpurun_out/ev3/bench_medium.json; cut -c1-300 gpurun_out/ev3/bench_medium.json
timeout 900 python bench.py --dtype f32 --steps 5 --warmup 2 --no-cpu-baseline --no-decode 2>/dev/null | tail -1 > gpurun_out/ev3/bench_f32.json; cut -c1-300 gpurun_out/ev3/bench_f32.json
MMTG_FORCE_DDP=1 timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-decode 2>/dev/null | tail -1 > gpurun_out/ev3/bench_forced_ddp.json
