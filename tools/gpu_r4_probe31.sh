#!/bin/bash
O=gpurun_out/p31; mkdir -p $O
( MMTG_EXTRA_DEFS=-DMMTG_CHAIN_SLEEP_X=2 python -m mmtg_amd.build --force --jobs 16 2>&1 | tail -1
  echo "== naps x2, polls 1.8 us apart"; timeout 300 python tools/decode_chain_timeline.py ) 2>&1 | grep -v amdgpu | tee $O/chain_timeline_sleep_ab.txt
