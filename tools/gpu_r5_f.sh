#!/bin/bash
# round 5: bf16x3 decode -- the three passes of a K slice as separate work items (MMTG_DECODE_X3_PASS_SPLIT=c_attn,attn.c_proj,mlp.c_proj,projector)
mkdir -p gpurun_out/r5f
run() { echo "== pass_split=$1 splits=$2 psplits=$3" | tee -a gpurun_out/r5f/pass_split_ab.txt
MMTG_DECODE_X3_PASS_SPLIT=$1 MMTG_DECODE_SPLITS=$2 MMTG_DECODE_X3_PSPLITS=$3 timeout 600 python3 bench.py --mode decode --dtype bf16x3 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['us_per_token_step'])" | tee -a gpurun_out/r5f/pass_split_ab.txt; }
run 0,0,0,0 2,4,1,8 8,2
run 1,0,0,0 2,4,1,8 8,2
run 1,0,0,0 1,4,1,8 8,2
run 0,1,0,0 2,2,1,8 8,2
run 0,1,0,0 2,4,1,8 8,2
run 0,0,1,0 2,4,1,4 8,2
run 0,0,1,0 2,4,1,2 8,2
run 0,0,0,1 2,4,1,8 4,1
run 1,1,1,1 1,4,1,4 4,1
run 1,1,1,1 2,2,1,4 4,1
MMTG_DECODE_X3_PASS_SPLIT=1,1,1,1 MMTG_DECODE_SPLITS=1,4,1,4 MMTG_DECODE_X3_PSPLITS=4,1 timeout 900 python -m pytest tests/test_decode_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "bf16x3" 2>&1 | tail -5 | tee gpurun_out/r5f/tests_pass_split.txt
