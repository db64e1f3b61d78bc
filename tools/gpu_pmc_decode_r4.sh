#!/bin/bash
# Round 4: HBM traffic of the FUSED decode token step (the shipped default) from PMC counters, at a generation length the profiler
# survives: bash tools/gpu_pmc_decode_r4.sh [decode_len=64].  Same method as tools/gpu_pmc_decode.sh (FETCH_SIZE and WRITE_SIZE in
# separate passes, --kernel-trace only, program directly after `--`, eager launches so every dispatch is visible; reads x2 on gfx950).
DLEN=${1:-64}
mkdir -p gpurun_out/pmc_decode_r4
export TMPDIR=/tmp
R=$(pwd)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_decode_r4 -o $c -- python3 bench.py --mode decode --decode-len $DLEN --steps 1 --warmup 1 --no-roofline --no-cpu-baseline --decode-eager > gpurun_out/pmc_decode_r4/$c.log 2>&1
  echo "pass $c rc=$?" >> gpurun_out/pmc_decode_r4/rc.txt
done
DLEN=$DLEN python3 - <<'PY'
import csv, glob, json, os, sys
sys.path.insert(0, ".")
from mmtg_amd import hip
DLEN = int(os.environ["DLEN"])
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc_decode_r4/%s_counter_collection.csv" % c)
    if not f:
        print("no counter file for", c); sys.exit(0)
    n, tot, per = 0, 0.0, {}
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c:
            continue
        name = r["Kernel_Name"]
        if not any(k in name for k in ("decode_", "gemm_dma_kernel", "splitk_finish", "ln_fwd_kernel")):
            continue
        if "ln_fold" in name:
            continue                  # once per generation, not part of the token step
        n += 1
        v = float(r["Counter_Value"])
        tot += v
        key = name.split("(")[0][-60:]
        per[key] = per.get(key, 0.0) + v
    out[c] = (n, tot, per)
gens, steps = 2, 15 + DLEN            # 1 warm-up + 1 timed generation, P + L token steps each
read_b = out["FETCH_SIZE"][1] * 1024 * 2 / (gens * steps)
write_b = out["WRITE_SIZE"][1] * 1024 / (gens * steps)
# algorithmic bytes of a token step at this length (bench.py decode_roofline's formula)
L, D, V, E, H, B = 12, 768, 13317, 2048, 512, 256
w_bytes = 2 * (L * 12 * D * D + V * D + E * H + H * D)
kv_row = 2 * L * D * 2
alg = w_bytes + B * kv_row * ((steps + 1) / 2.0 + 1)
res = {"what": "decode token step, batch 256, every kernel of the FUSED step (the shipped default; eager launches of the graph's node list)",
       "step": "fused", "decode_len": DLEN,
       "kernel_source_sha": hip.source_sha(), "dispatches_counted": out["FETCH_SIZE"][0],
       "hbm_read_bytes_per_token_step": round(read_b), "hbm_write_bytes_per_token_step": round(write_b),
       "hbm_bytes_per_token_step": round(read_b + write_b), "algorithmic_bytes_per_token_step_at_this_length": round(alg),
       "read_bytes_per_token_step_by_kernel": {k: round(v * 2048 / (gens * steps)) for k, v in sorted(out["FETCH_SIZE"][2].items(), key=lambda kv: -kv[1])},
       "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over python3 bench.py --mode decode --decode-len %d --steps 1 --warmup 1 --decode-eager; KB units; reads x2 (gfx950 FETCH_SIZE counts 128-B requests at 64 B); averaged over both generations' %d token steps" % (DLEN, steps)}
json.dump(res, open("gpurun_out/decode_pmc_traffic_fused_len%d.json" % DLEN, "w"), indent=1)
print(json.dumps(res))
PY
find gpurun_out/pmc_decode_r4 -name "*kernel_trace.csv" -size +4M -delete
find gpurun_out/pmc_decode_r4 -name "*counter_collection.csv" -size +4M -delete
