#!/bin/bash
# HBM traffic of the dominant kernel (GEMM) with PMC counters, separate passes as the guide prescribes
mkdir -p gpurun_out/pmc2
export TMPDIR=/tmp
R=$(pwd)
PMC=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc2 -o fetch -- python3 tools/bench_gemm.py > gpurun_out/pmc2/log1.txt 2>&1
PMC=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc2 -o write -- python3 tools/bench_gemm.py > gpurun_out/pmc2/log2.txt 2>&1
python3 - <<'PY'
import csv, glob, collections
for tag in ("fetch", "write"):
    f = glob.glob('gpurun_out/pmc2/%s_counter_collection.csv' % tag)
    if not f: print("no", tag); continue
    for r in csv.DictReader(open(f[0])):
        if 'gemm' in r['Kernel_Name']:
            print(tag, r['Dispatch_Id'], r['Kernel_Name'][40:100], r['Counter_Name'], r['Counter_Value'])
PY
