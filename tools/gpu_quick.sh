#!/bin/bash
# scratch: split-K decode sweep
for sp in "2,3,2,6" "2,2,2,4" "1,2,1,4" "2,3,1,6" "1,3,1,6" "2,4,2,6" "2,3,2,8" "1,1,1,2"; do echo "== splits $sp"; MMTG_DECODE_SPLITS=$sp timeout 600 python bench.py --mode decode --no-cpu-baseline 2>&1 | tail -1 | grep -o '"us_per_token_step": [0-9.]*'; done
