#!/bin/bash
for i in 1 2 3; do
echo "== plain"; MMTG_NO_FEW_ROWS=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo
echo "== split"; timeout 600 python bench.py --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo
done
