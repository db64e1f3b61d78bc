#!/bin/bash
mkdir -p gpurun_out/p7
O=gpurun_out/p7
( echo "== 3 workgroups per CU"; timeout 300 python tools/decode_persist_timeline.py
  echo "== 2 workgroups per CU"; MMTG_DECODE_PERSIST_WGS=2 timeout 300 python tools/decode_persist_timeline.py ) 2>&1 | grep -v amdgpu > $O/persist_timeline.txt
cat $O/persist_timeline.txt
