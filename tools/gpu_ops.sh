#!/bin/bash
# Runs the per-kernel parity tests on the GPU box; full report (no -x) into gpurun_out/.
mkdir -p gpurun_out
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" || true
timeout 1500 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider "$@" > gpurun_out/ops.log 2>&1
echo "exit $?" >> gpurun_out/ops.log
tail -60 gpurun_out/ops.log
