#!/bin/bash
mkdir -p gpurun_out/pm
export TMPDIR=/tmp
R=$(pwd)
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pm -o P -- python3 -m pytest tests/test_ops_gpu.py -x -q --no-header -p no:cacheprovider -k "decode_gemm" > gpurun_out/pm/P.log 2>&1
echo "op test under PMC rc=$?"; grep -v "^W2026\|^I2026" gpurun_out/pm/P.log | tail -5
for L in 32 64 96; do
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pm -o L$L -- python3 bench.py --mode decode --steps 1 --warmup 1 --no-roofline --no-cpu-baseline --decode-eager --decode-len $L > gpurun_out/pm/L$L.log 2>&1
echo "len $L rc=$?"
done
find gpurun_out/pm -name "*.csv" -size +1M -delete
