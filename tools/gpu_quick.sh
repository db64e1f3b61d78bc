#!/bin/bash
for st in "0,0,0" "256,512,1" "256,512,2" "256,512,3" "0,512,2"; do echo "== stagger $st"; MMTG_STAGGER=$st KSWEEP=1 timeout 300 python tools/bench_gemm.py 2>&1 | grep "N=3072 K-sweep" | grep "K=   768\|K=  3072"; MMTG_STAGGER=$st timeout 100 python tools/gemm_timeline.py 15104 3072 768 NT 0 128 2>&1 | grep "span\|starts"; done
