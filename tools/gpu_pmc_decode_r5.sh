#!/bin/bash
# Round 5: HBM traffic of the decode token step from PMC counters AT THE BENCHMARKED LENGTH (128 positions).
#   bash tools/gpu_pmc_decode_r5.sh [decode_len=128] [dtype=bf16]
# What changed against tools/gpu_pmc_decode_r4.sh (whose counter pass hung / crashed at 128 positions, 18.9 k dispatches):
#   * ONE generation (--warmup 0 --steps 1 with --decode-eager: no warm-up generation) = len x ~66 dispatches, 8.4 k at 128
#     (the 15 prompt positions are prefilled by training-side kernels, which the filter below leaves out);
#   * --kernel-include-regex decode_ : only the token step's kernels are instrumented (no torch fills / copies / once-per-generation folds).
# FETCH_SIZE and WRITE_SIZE in separate passes, --kernel-trace only, the program directly after `--` (MI355X_MICROARCH.md).
DLEN=${1:-128}
DT=${2:-bf16}
OUT=gpurun_out/pmc_decode_r5_$DT
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
# (the counter pass of this workload hangs INTERMITTENTLY inside rocprofv3 -- rounds 3-5: same command, same sources, 2 of 3 attempts complete in ~25 s,
#  the third sits until the timeout -- so every pass gets up to three attempts under a short timeout; rc.txt records each one)
rm -f $R/$OUT/rc.txt
for c in FETCH_SIZE WRITE_SIZE; do
  for attempt in 1 2 3; do
    rm -f $R/$OUT/${c}_counter_collection.csv
    timeout 240 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "decode_" --output-format csv -d $R/$OUT -o $c -- python3 $R/bench.py --mode decode --dtype $DT --decode-len $DLEN --steps 1 --warmup 0 --no-roofline --no-cpu-baseline --decode-eager > $R/$OUT/$c.log 2>&1
    rc=$?
    echo "pass $c attempt $attempt rc=$rc" >> $R/$OUT/rc.txt
    [ $rc -eq 0 ] && [ -s $R/$OUT/${c}_counter_collection.csv ] && break
  done
done
cd $R
cat $OUT/rc.txt
DLEN=$DLEN DT=$DT OUT=$OUT python3 - <<'PY'
import csv, glob, json, os, sys
sys.path.insert(0, ".")
from mmtg_amd import hip
DLEN, DT, OUT = int(os.environ["DLEN"]), os.environ["DT"], os.environ["OUT"]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("%s/%s_counter_collection.csv" % (OUT, c))
    if not f:
        print("no counter file for", c); sys.exit(0)
    n, tot, per = 0, 0.0, {}
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c:
            continue
        name = r["Kernel_Name"]
        if "decode_" not in name:
            continue
        n += 1
        v = float(r["Counter_Value"])
        tot += v
        key = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-60:]
        per[key] = per.get(key, 0.0) + v
    out[c] = (n, tot, per)
# the prompt is prefilled in one batched pass (training-side kernels: not matched by the decode_ filter), the token steps run at
# positions 15 .. 15 + DLEN - 1 (MMTG_DECODE_PREFILL=0: the prompt as 15 more token steps from position 0)
first = 0 if os.environ.get("MMTG_DECODE_PREFILL") == "0" else 15
gens, steps = 1, 15 + DLEN - first
read_b = out["FETCH_SIZE"][1] * 1024 * 2 / (gens * steps)
write_b = out["WRITE_SIZE"][1] * 1024 / (gens * steps)
L, D, V, E, H, B = 12, 768, 13317, 2048, 512, 256
esz = 2 if DT == "bf16" else 4
w_bytes = esz * (L * 12 * D * D + V * D + E * H + H * D)
kv_row = 2 * L * D * esz
alg = w_bytes + B * kv_row * (first + (steps - 1) / 2.0 + 1.0)      # bench.py decode_mean_kv_rows
res = {"what": "decode token step, batch 256, dtype %s, the kernels of the fused token step (eager launches of the graph's node list)" % DT,
       "step": "fused" if DT == "bf16" else DT, "dtype": DT, "decode_len": DLEN,
       "kernel_source_sha": hip.source_sha(), "dispatches_counted": out["FETCH_SIZE"][0],
       "hbm_read_bytes_per_token_step": round(read_b), "hbm_write_bytes_per_token_step": round(write_b),
       "hbm_bytes_per_token_step": round(read_b + write_b), "algorithmic_bytes_per_token_step_at_this_length": round(alg),
       "traffic_over_algorithmic": round((read_b + write_b) / alg, 3),
       "read_bytes_per_token_step_by_kernel": {k: round(v * 2048 / (gens * steps)) for k, v in sorted(out["FETCH_SIZE"][2].items(), key=lambda kv: -kv[1])},
       "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) --kernel-include-regex decode_ over python3 bench.py --mode decode "
                 "--dtype %s --decode-len %d --steps 1 --warmup 0 --decode-eager (ONE generation, %d token steps); KB units; reads x2 (gfx950 FETCH_SIZE "
                 "counts 128-B requests at 64 B)" % (DT, DLEN, steps)}
json.dump(res, open("gpurun_out/decode_pmc_traffic_%s_len%d.json" % (DT, DLEN), "w"), indent=1)
print(json.dumps(res))
PY
find $OUT -name "*kernel_trace.csv" -size +4M -delete
find $OUT -name "*counter_collection.csv" -size +4M -delete
