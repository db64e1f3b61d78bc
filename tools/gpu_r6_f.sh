#!/bin/bash
# Round 6, call F: the default bench line on the split bench.py (all objects), 3000-step memorisation curves of the three fast modes.
mkdir -p gpurun_out/r6f
E=gpurun_out/r6f
timeout 1200 python bench.py > $E/bench_default.json 2> $E/bench_default.err; tail -c 600 $E/bench_default.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6f/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: d["roofline"].get(k) for k in ("frac", "traffic", "decode_tokens_per_s", "decode_us_per_token_step", "decode_frac", "parity_train_tokens_per_s", "forward_parity_train_tokens_per_s", "conditioning_unfused_f32_frac_hbm")})
print({k: (v.get("error") if isinstance(v, dict) and "error" in v else "ok") for k, v in d.items() if isinstance(v, dict)})
print(json.dumps(d.get("conditioning", {}).get("verdict")), json.dumps({k: (v.get("frac_hbm"), v.get("us")) for k, v in d.get("conditioning", {}).items() if isinstance(v, dict) and "frac_hbm" in v}))
PY
for m in bf16x3f bf16x3 bf16; do MODE=$m timeout 400 python tools/train_curve.py 3000 2>&1 | grep -v amdgpu.ids | tee $E/train_curve_3000_$m.txt | tail -4; done
