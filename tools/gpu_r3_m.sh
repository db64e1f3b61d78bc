#!/bin/bash
mkdir -p gpurun_out/r3m
timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_ops_gpu.py -x -q --no-header -p no:cacheprovider -k "decode or logits_process or kv_cache or generate or sample" 2>&1 | tail -3 | tee gpurun_out/r3m/pytest_decode.txt
rm -f gpurun_out/r3k/decode_ab.txt
bash tools/gpu_r3_k.sh; cp gpurun_out/r3k/decode_ab.txt gpurun_out/r3m/decode_ab.txt
bash tools/gpu_r3_l.sh > /dev/null 2>&1
python3 - <<'PY'
import csv
for t in ("base","new"):
    rows=list(csv.DictReader(open('gpurun_out/r3l/%s_kernel_stats.csv'%t)))
    for r in rows:
        if any(k in r['Name'] for k in ("decode_attn","decode_select","splitk_finish","gemm_dma_kernel<false, false, 64")):
            print(t, r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,2), "min", round(float(r['MinNs'])/1e3,2), "max", round(float(r['MaxNs'])/1e3,2))
PY
