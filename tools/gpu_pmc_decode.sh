#!/bin/bash
# HBM traffic of the decode token step (every kernel of a generation: split-K products, KV-cache attention, finish kernels,
# selection) from PMC counters: FETCH_SIZE and WRITE_SIZE in separate passes, --kernel-trace only, the program directly after
# `--` (MI355X_MICROARCH.md; gfx950 FETCH_SIZE counts 128-B requests at 64 B -> reads x2).  The generation runs WITHOUT graph
# capture in these passes (same launches, eager) so that every dispatch is visible to the counter collection.
# Writes profiles-ready gpurun_out/decode_pmc_traffic.json (bench.py picks up profiles/r*_decode_pmc_traffic.json by source sha).
mkdir -p gpurun_out/pmc_decode
export TMPDIR=/tmp
R=$(pwd)
# NOTE (round 3): with the FUSED token step the counter passes die at --decode-len 128 with a host-side segfault inside the
# profiler's launch interception (first mmtg_decode_gemm mode-2 launch; lengths <= 96, the op test of that kernel and every
# non-counter run are fine), so the passes measure the round-2 step (MMTG_DECODE_FUSED=0: products + finish launches).  Every
# pass runs under `timeout`: a counter pass that hangs must not hold the GPU box.
export MMTG_DECODE_FUSED=0
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_decode -o $c -- python3 bench.py --mode decode --steps 1 --warmup 1 --no-roofline --no-cpu-baseline --decode-eager > gpurun_out/pmc_decode/$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, json, sys
sys.path.insert(0, ".")
from mmtg_amd import hip
out = {}
skip = ("at::native", "__amd_rocclr", "cast_from_f32", "transpose_batch", "gru_cell", "alpha_fwd", "beta_fuse", "ln_fwd3", "Cijk", "fill")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc_decode/%s_counter_collection.csv" % c)
    n, tot, per = 0, 0.0, {}
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c:
            continue
        name = r["Kernel_Name"]
        if not any(k in name for k in ("decode_", "gemm_dma_kernel", "splitk_finish", "ln_fwd_kernel")):
            continue
        n += 1
        v = float(r["Counter_Value"])
        tot += v
        key = name.split("(")[0][-60:]
        per[key] = per.get(key, 0.0) + v
    out[c] = (n, tot, per)
# two generations ran (1 warm-up + 1 timed), each P + L = 15 + 128 token steps
gens, steps = 2, 143
read_b = out["FETCH_SIZE"][1] * 1024 * 2 / (gens * steps)
write_b = out["WRITE_SIZE"][1] * 1024 / (gens * steps)
res = {"what": "decode token step, batch 256, every kernel of the ROUND-2 (unfused, MMTG_DECODE_FUSED=0) step (eager launches of the graph's node list)",
       "step": "unfused",
       "kernel_source_sha": hip.source_sha(), "dispatches_counted": out["FETCH_SIZE"][0],
       "hbm_read_bytes_per_token_step": round(read_b), "hbm_write_bytes_per_token_step": round(write_b),
       "hbm_bytes_per_token_step": round(read_b + write_b),
       "read_bytes_per_token_step_by_kernel": {k: round(v * 2048 / (gens * steps)) for k, v in sorted(out["FETCH_SIZE"][2].items(), key=lambda kv: -kv[1])},
       "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over python3 bench.py --mode decode --steps 1 --warmup 1 --decode-eager; KB units; reads x2 (gfx950 FETCH_SIZE counts 128-B requests at 64 B); averaged over both generations' 143 token steps"}
json.dump(res, open("gpurun_out/decode_pmc_traffic.json", "w"), indent=1)
print(json.dumps(res))
PY
find gpurun_out/pmc_decode -name "*kernel_trace.csv" -size +4M -delete
find gpurun_out/pmc_decode -name "*counter_collection.csv" -size +4M -delete
