#!/bin/bash
# round 5: bf16x3 decode -- ring depth 8 (one workgroup per CU) against 4 (two per CU), K-split sweeps
mkdir -p gpurun_out/r5e
run() { echo "== nbuf=$1 splits=$2 psplits=$3" | tee -a gpurun_out/r5e/nbuf_ab.txt
MMTG_DECODE_X3_NBUF=$1 MMTG_DECODE_SPLITS=$2 MMTG_DECODE_X3_PSPLITS=$3 timeout 600 python3 bench.py --mode decode --dtype bf16x3 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['us_per_token_step'])" | tee -a gpurun_out/r5e/nbuf_ab.txt; }
run 4 2,4,1,8 8,2
run 8 2,4,1,8 8,2
run 8 1,4,1,4 8,2
run 8 1,2,1,4 4,1
run 8 1,4,1,5 8,2
run 4 2,4,1,8 4,1
run 4 2,4,1,8 8,1
