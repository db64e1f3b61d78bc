#!/bin/bash
mkdir -p gpurun_out/r3o
MMTG_FORCE_DDP=1 timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-decode 2> gpurun_out/r3o/forced_ddp.err | tail -1 > gpurun_out/r3o/bench_forced_ddp.json; python3 -c "
import json; d=json.loads(open('gpurun_out/r3o/bench_forced_ddp.json').read()); print(d['value'], d['ms_per_step'], d.get('ddp'), d['config']['parallelism'])"
tail -2 gpurun_out/r3o/forced_ddp.err
timeout 900 python bench.py 2> gpurun_out/r3o/bench_default.err | tail -1 > gpurun_out/r3o/bench_default.json; python3 -c "
import json; d=json.loads(open('gpurun_out/r3o/bench_default.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['per_category_ms_per_step']); print(d['decode']['value'], d['f32']['train']['value'], d['f32']['decode']['value'], d['conditioning']['GB/s'])"
