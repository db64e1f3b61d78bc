#!/bin/bash
# same-box A/B of the decode bench: base tree (gpurun_ab/base) vs working tree, interleaved twice
mkdir -p gpurun_out/r3k
for rep in 1 2; do
  for t in base new; do
    if [ $t = base ]; then d=gpurun_ab/base; else d=.; fi
    (cd $d && timeout 600 python bench.py --mode decode --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1) | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('$t', d['value'], d['config']['us_per_token_step'], r['per_category_ms_per_generation'])" | tee -a gpurun_out/r3k/decode_ab.txt
  done
done
