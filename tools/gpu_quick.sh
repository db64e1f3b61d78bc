#!/bin/bash
# scratch: LayerNorm backward v2 block-count sweep
for cap in 1024 512 256; do echo "== cap $cap"; MMTG_LN_CAP=$cap timeout 300 python tools/bench_rowops.py 2>&1 | grep "ln bwd"; done
