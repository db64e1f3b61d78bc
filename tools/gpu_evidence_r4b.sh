#!/bin/bash
# Round 4 evidence, pass 2 (after the pass-1 PMC json files were copied into profiles/): the bench lines
mkdir -p gpurun_out/ev4
E=gpurun_out/ev4
timeout 1200 python bench.py > $E/bench_default.json 2> $E/bench_default.err; tail -c 600 $E/bench_default.json
timeout 600 python bench.py --mode decode > $E/bench_decode.json 2> $E/bench_decode.err; tail -c 300 $E/bench_decode.json
timeout 600 python bench.py --config medium --no-cpu-baseline > $E/bench_medium.json 2> $E/bench_medium.err; tail -c 300 $E/bench_medium.json
MMTG_FORCE_DDP=1 timeout 600 python bench.py --no-cpu-baseline --no-decode --steps 10 --warmup 3 > $E/bench_forced_ddp_world1.json 2> $E/bench_forced_ddp.err; tail -c 700 $E/bench_forced_ddp_world1.json
