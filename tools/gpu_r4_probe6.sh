#!/bin/bash
mkdir -p gpurun_out/p6
O=gpurun_out/p6
( echo "== B=64 L=2"; B=64 LAYERS=2 LEN=120 timeout 300 python tools/persist_debug.py
  echo "== B=256 L=12"; B=256 LAYERS=12 LEN=120 timeout 600 python tools/persist_debug.py ) 2>&1 | grep -v amdgpu > $O/persist_debug_un16.txt
cat $O/persist_debug_un16.txt
