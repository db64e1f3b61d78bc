#!/bin/bash
for v in 1 0; do echo "== attn split $v"; MMTG_DECODE_ATTN_SPLIT=$v timeout 600 python bench.py --mode decode --no-cpu-baseline 2>&1 | tail -1 | grep -o '"us_per_token_step": [0-9.]*'; done
