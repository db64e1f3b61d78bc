#!/bin/bash
mkdir -p gpurun_out/p14
rm -rf gpurun_out/pmc_decode_r4
bash tools/gpu_pmc_decode_r4.sh 128 > gpurun_out/p14/pmc_decode_128.log 2>&1
cat gpurun_out/pmc_decode_r4/rc.txt; tail -2 gpurun_out/p14/pmc_decode_128.log | cut -c1-700
tail -5 gpurun_out/pmc_decode_r4/FETCH_SIZE.log | cut -c1-300
cp gpurun_out/decode_pmc_traffic_fused_len128.json gpurun_out/p14/ 2>/dev/null
