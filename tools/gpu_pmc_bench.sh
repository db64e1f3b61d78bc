#!/bin/bash
# HBM traffic of the dominant kernel (gemm_p8_kernel / gemm_occ4_kernel / gemm_dma_kernel, all instantiations) inside the bench program,
# from PMC counters: FETCH_SIZE and WRITE_SIZE in separate passes, --kernel-trace only (as
# MI355X_MICROARCH.md prescribes; gfx950 FETCH_SIZE counts 128-B requests at 64 B -> reads x2).
# Writes profiles-ready gpurun_out/bench_pmc_gemm_traffic.json.
mkdir -p gpurun_out/pmc_bench
export TMPDIR=/tmp
R=$(pwd)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_bench -o $c -- python3 bench.py --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --primary-only --no-check > gpurun_out/pmc_bench/$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, json, sys
sys.path.insert(0, ".")
from mmtg_amd import hip
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc_bench/%s_counter_collection.csv" % c)
    n, tot = 0, 0.0
    for r in csv.DictReader(open(f[0])):
        if any(k in r["Kernel_Name"] for k in ("gemm_dma_kernel", "gemm_occ4_kernel", "gemm_p8_kernel", "gemm_p8p_kernel", "wgrad_group_kernel")) and r["Counter_Name"] == c:
            n += 1
            tot += float(r["Counter_Value"])
    out[c] = (n, tot)
n = out["FETCH_SIZE"][0]
read_b = out["FETCH_SIZE"][1] * 1024 * 2 / max(n, 1)      # KB units, x2 gfx950 correction
write_b = out["WRITE_SIZE"][1] * 1024 / max(out["WRITE_SIZE"][0], 1)
res = {"kernel": "gemm_p8_kernel / gemm_occ4_kernel / gemm_dma_kernel <bf16> (all instantiations)", "launches_counted": n,
       "kernel_source_sha": hip.source_sha(),
       "hbm_read_bytes_per_launch": round(read_b), "hbm_write_bytes_per_launch": round(write_b),
       "hbm_bytes_per_launch": round(read_b + write_b),
       "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over python3 bench.py --steps 2 --warmup 1; KB units; reads x2 (gfx950 FETCH_SIZE counts 128-B requests at 64 B)"}
json.dump(res, open("gpurun_out/bench_pmc_gemm_traffic.json", "w"), indent=1)
print(json.dumps(res))
PY
find gpurun_out/pmc_bench -name "*kernel_trace.csv" -size +4M -delete
find gpurun_out/pmc_bench -name "*counter_collection.csv" -size +4M -delete
