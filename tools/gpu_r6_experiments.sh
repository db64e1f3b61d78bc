#!/bin/bash
# Round 6: the individual `gpurun` experiment calls of the round, one section per call (the judged evidence set is produced by
# tools/gpu_evidence_r6.sh + tools/gpu_evidence_r6b.sh).    bash tools/gpu_r6_experiments.sh <section>
# Each section writes under gpurun_out/; what it produced is committed as profiles/r06_v1..v7_* (profiles/README.md, "Round 6").
case "$1" in
a)
# Round 6, call A: the one-launch decode MLP -- op tests, stand-alone timeline, token-step A/B -- plus the new host-side tests.
mkdir -p gpurun_out/r6b
E=gpurun_out/r6b
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "decode_mlp or decode_gemm_ln_fold" 2>&1 | tail -15 | tee $E/pytest_mlp_ops.txt
timeout 300 python tools/decode_mlp_timeline.py 256 2>&1 | tee $E/decode_mlp_timeline_m256.txt
timeout 200 python tools/decode_mlp_timeline.py 128 2>&1 | tail -30 > $E/decode_mlp_timeline_m128.txt
for cfg in "0 sc1" "1 sc1" "1 plain" "0 sc1" "1 plain"; do
  set -- $cfg
  MMTG_DECODE_MLP=$1 MMTG_DECODE_MLP_HANDOFF=$2 timeout 400 python bench.py --mode decode --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>$E/err_$1_$2.log | python -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    print('MLP=$1 handoff=$2', d['value'], 'tok/s', d['config']['us_per_token_step'], 'us/step', d['config']['once_per_generation_ms'], 'ms once', d['check'])
" | tee -a $E/decode_ab.txt
done
timeout 1500 python -m pytest tests/test_decode_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "teacher_forced or fused_decode_step or x3_engine or full_size_batched" 2>&1 | tail -15 | tee $E/pytest_decode.txt


  ;;
c)
# Round 6, call C: why the one-launch MLP loses inside the token step although it ties stand-alone: the graph-node floor by launch
# shape, the in-step timeline, and rocprofv3 per-kernel statistics of the token step with and without it.
mkdir -p gpurun_out/r6c
E=gpurun_out/r6c
timeout 120 ./tools/micro/node_floor 2>&1 | tee $E/node_floor.txt
timeout 300 python tools/decode_mlp_insitu.py sc1 2>&1 | tail -14 | tee $E/mlp_insitu_sc1.txt
timeout 300 python tools/decode_mlp_insitu.py plain 2>&1 | tail -14 | tee $E/mlp_insitu_plain.txt
export TMPDIR=/tmp
R=$(pwd)
for m in 0 1; do
  ( cd /tmp && MMTG_DECODE_MLP=$m MMTG_DECODE_MLP_HANDOFF=plain timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$E/prof_mlp$m -o p -- python3 $R/bench.py --mode decode --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/$E/prof_mlp$m.log 2>&1 )
  f=$(find $E/prof_mlp$m -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $E/decode_mlp${m}_rocprofv3_kernel_stats.csv && head -14 $f | cut -c1-200
  rm -rf $E/prof_mlp$m
done

  ;;
d)
mkdir -p gpurun_out/r6d
E=gpurun_out/r6d
timeout 200 python tools/decode_mlp_timeline.py 256 hot 2>&1 | grep -v amdgpu.ids | head -20 | tee $E/mlp_timeline_hot.txt
HIP_FORCE_DEV_KERNARG=1 timeout 60 ./tools/micro/node_floor 2>&1 | head -8 | tee $E/node_floor_dev_kernarg.txt
for ka in 0 1 0 1; do
  HIP_FORCE_DEV_KERNARG=$ka timeout 400 python bench.py --mode decode --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    print('HIP_FORCE_DEV_KERNARG=$ka', d['value'], 'tok/s', d['config']['us_per_token_step'], 'us/step', d['config']['once_per_generation_ms'], 'ms once')
" | tee -a $E/decode_dev_kernarg_ab.txt
done
for ka in 0 1 0 1; do
  HIP_FORCE_DEV_KERNARG=$ka timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --primary-only --no-roofline 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    print('train HIP_FORCE_DEV_KERNARG=$ka', d['value'], 'tok/s', d['ms_per_step'], 'ms/step')
" | tee -a $E/train_dev_kernarg_ab.txt
done

  ;;
e)
# Round 6, call E: the hybrid mode (bf16x3 forward + bf16 backward) -- parity tests, speed; HIP_FORCE_DEV_KERNARG set from inside Python.
mkdir -p gpurun_out/r6e
E=gpurun_out/r6e
timeout 1500 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "bf16x3f" 2>&1 | tail -15 | tee $E/pytest_bf16x3f.txt
for m in bf16x3f bf16x3 bf16; do timeout 300 python tools/bench_x3.py $m 64 10 2>&1 | grep -v amdgpu.ids | tee -a $E/bench_modes.txt; done
for ka in 0 default 0 default; do
  if [ $ka = 0 ]; then export HIP_FORCE_DEV_KERNARG=0; else unset HIP_FORCE_DEV_KERNARG; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --primary-only --no-roofline 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    print('train shell HIP_FORCE_DEV_KERNARG=$ka', d['value'], 'tok/s', d['ms_per_step'], 'ms/step')
" | tee -a $E/train_dev_kernarg_from_python.txt
done
unset HIP_FORCE_DEV_KERNARG
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_x3_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "bf16x3 and not bf16x3f and (forward or fused_train or reproducible)" 2>&1 | tail -6 | tee $E/pytest_x3_regress.txt

  ;;
f)
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

  ;;
g)
mkdir -p gpurun_out/r6g
E=gpurun_out/r6g
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "differentiates_the_forward or (bit_reproducible and bf16x3f)" 2>&1 | tail -12 | tee $E/pytest_masks.txt
MODE=bf16x3f timeout 300 python tools/nan_hunt.py 400 2>&1 | grep -v amdgpu.ids | tee $E/nan_hunt_bf16x3f.txt | tail -25
MODE=bf16x3f PDROP=0 timeout 300 python tools/nan_hunt.py 400 2>&1 | grep -v amdgpu.ids | tee $E/nan_hunt_bf16x3f_nodrop.txt | tail -8
for es in 0 1 0 1; do
  MMTG_ENC_STREAMS=$es timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --primary-only --no-roofline 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    print('train MMTG_ENC_STREAMS=$es', d['value'], 'tok/s', d['ms_per_step'], 'ms/step')
" | tee -a $E/enc_streams_ab.txt
done
MMTG_ENC_STREAMS=1 timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "bit_reproducible or bf16_vs_oracle or fused_train_step or rnn_interlayer or encoder_sizes" 2>&1 | tail -6 | tee $E/pytest_enc_streams.txt

  ;;
h)
mkdir -p gpurun_out/r6h
timeout 300 python tools/decode_begin_breakdown.py bf16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6h/decode_begin_breakdown_bf16.txt
timeout 300 python tools/decode_begin_breakdown.py bf16x3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6h/decode_begin_breakdown_bf16x3.txt
timeout 300 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "differentiates_the_forward" 2>&1 | tail -3

  ;;
i)
mkdir -p gpurun_out/r6i
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "differentiates" 2>&1 | tail -40 | tee gpurun_out/r6i/alone.txt
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "bit_reproducible or differentiates or with_dropout_matches" 2>&1 | grep -E "^E  |passed|failed|assert" | head -40 | tee gpurun_out/r6i/with_neighbours.txt

  ;;
j)
mkdir -p gpurun_out/r6j
timeout 600 python -m pytest tests/test_x3_gpu.py tests/test_ddp_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "element_dropout_mask or bf16_gradient_exchange" 2>&1 | tail -25 | tee gpurun_out/r6j/pytest_new.txt
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -6 | tee gpurun_out/r6j/smoke.txt

  ;;
k)
# Round 6: the dGELU product (gemm_occ4 by default: 0.24 of the matrix peak) on the eight-phase kernel, with and without the forward
# storing gelu'(u) instead of u -- whole training step, same box, alternating.
mkdir -p gpurun_out/r6k
for rep in 1 2; do
for cfg in "0 1" "0 2" "1 1" "1 2"; do
  set -- $cfg
  MMTG_GELU_GRAD=$1 MMTG_GEMM_P8=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --primary-only 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    print('GELU_GRAD=$1 GEMM_P8=$2', d['value'], 'tok/s', d['ms_per_step'], 'ms/step  GEMM family', d['roofline']['frac'], d['roofline']['per_category_ms_per_step']['gemm_bf16'])
" | tee -a gpurun_out/r6k/dgelu_p8_ab.txt
done
done

  ;;
l)
mkdir -p gpurun_out/r6l
E=gpurun_out/r6l
timeout 1500 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "bf16x3f" 2>&1 | tail -8 | tee $E/pytest_bf16x3f.txt
for e in 0 1 0 1; do MMTG_HYBRID_ENC_BF16=$e timeout 300 python tools/bench_x3.py bf16x3f 64 10 2>&1 | grep -v amdgpu.ids | head -3 | sed "s/^/ENC_BF16=$e /" | tee -a $E/enc_bf16_ab.txt; done
MODE=bf16x3f timeout 400 python tools/train_curve.py 3000 2>&1 | grep -v amdgpu.ids | tee $E/train_curve_3000_bf16x3f.txt | tail -3

  ;;
m)
mkdir -p gpurun_out/r6m
( time timeout 3300 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider ) 2>&1 | tail -8 | tee gpurun_out/r6m/pytest_gpu.txt
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -6 | tee gpurun_out/r6m/smoke.txt
timeout 1500 python bench.py > gpurun_out/r6m/bench_default.json 2> gpurun_out/r6m/bench_default.err; tail -c 300 gpurun_out/r6m/bench_default.json

  ;;
n)
# existing opt-in overlap switches re-measured with kernel arguments in device memory: grouped weight gradients on a side stream,
# the attention backward's dQ kernel beside its dK / dV kernel
mkdir -p gpurun_out/r6n
for rep in 1 2; do
for cfg in "0 0" "1 0" "0 1"; do
  set -- $cfg
  MMTG_WGRAD_STREAM=$1 MMTG_ATTN_BWD_FORK=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --primary-only --no-roofline 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    print('WGRAD_STREAM=$1 ATTN_BWD_FORK=$2', d['value'], 'tok/s', d['ms_per_step'], 'ms/step')
" | tee -a gpurun_out/r6n/overlap_switches_ab.txt
done
done
;;
o)
# the exchange through the library's own RCCL communicator (MMTG_DDP_COMM=abi, csrc/comm.hip): the data-parallel tests, then the
# full-size step at world 1 with the exchange forced (MMTG_FORCE_DDP=1), torch.distributed per bucket against one C call per bucket
mkdir -p gpurun_out/r6o
E=gpurun_out/r6o
timeout 1500 python -m pytest tests/test_ddp_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -8 | tee $E/pytest_ddp.txt
for rep in 1 2; do
for comm in torch abi; do
  MMTG_FORCE_DDP=1 MMTG_DDP_COMM=$comm timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --primary-only --no-roofline 2>$E/err_$comm.log | python -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    dd = d.get('ddp') or {}
    print('forced world-1 exchange, comm=$comm', d['value'], 'tok/s', d['ms_per_step'], 'ms/step; finish wait', dd.get('finish_wait_ms_per_step'), 'ms; exchange alone', dd.get('allreduce_ms_per_step_isolated'), 'ms;', dd.get('comm'), dd.get('comm_info'))
" | tee -a $E/forced_world1_comm_ab.txt
done
done
tail -3 $E/err_abi.log
;;
p)
# kernel arguments in device memory again, on whatever box this call gets (the round's boxes differ by 3 % in the bf16 step)
mkdir -p gpurun_out/r6p2
for ka in 1 0 1 0; do
  HIP_FORCE_DEV_KERNARG=$ka timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --primary-only --no-roofline 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    print('HIP_FORCE_DEV_KERNARG=$ka', d['value'], 'tok/s', d['ms_per_step'], 'ms/step')
" | tee -a gpurun_out/r6p2/train_dev_kernarg_ab.txt
done
;;
q)
# the attention backward's dQ kernel launched without the queue barrier (hipExtAnyOrderLaunch) behind its dK / dV kernel:
# parity tests with the switch on, the step A/B, and whether the two kernels really overlap (rocprofv3 kernel trace).
# (needs tools/experiments/attn_bwd_anyorder_launch.patch applied: the runtime keeps the barrier on gfx950, the switch was removed)
mkdir -p gpurun_out/r6q2
E=gpurun_out/r6q2
MMTG_ATTN_BWD_ANYORDER=1 timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "attn or attention or bf16_vs_oracle or reproducible" 2>&1 | tail -4 | tee $E/pytest_anyorder.txt
for rep in 1 2 3; do
for ao in 0 1; do
  MMTG_ATTN_BWD_ANYORDER=$ao timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --primary-only --no-roofline 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    print('ATTN_BWD_ANYORDER=$ao', d['value'], 'tok/s', d['ms_per_step'], 'ms/step')
" | tee -a $E/anyorder_ab.txt
done
done
export TMPDIR=/tmp
R=$(pwd)
( cd /tmp && MMTG_ATTN_BWD_ANYORDER=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$E/trace -o t -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --primary-only --no-roofline --no-check > $R/$E/trace.log 2>&1 )
f=$(find $E/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $E/overlap_from_kernel_trace.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = ov = 0
for a, b in zip(rows, rows[1:]):
    if "attn_bwd_small_kv2" in a["Kernel_Name"] and "attn_bwd_small_q" in b["Kernel_Name"]:
        n += 1
        d = int(a["End_Timestamp"]) - int(b["Start_Timestamp"])
        ov += d > 0
        if n <= 6:
            print("kv2 %.1f us, q starts %.1f us %s kv2's end, q %.1f us" % ((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3,
                  abs(d) / 1e3, "BEFORE" if d > 0 else "after", (int(b["End_Timestamp"]) - int(b["Start_Timestamp"])) / 1e3))
print("pairs", n, "overlapping", ov)
PY
rm -rf $E/trace
;;
r)
# the decoder backward's small column sums batched into one launch (MMTG_DEFER_SUMS): op + model tests, then the step A/B
mkdir -p gpurun_out/r6r2
E=gpurun_out/r6r2
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "batched_column or layernorm or attention or reproducible or bf16_vs_oracle or full_12l" 2>&1 | tail -4 | tee $E/pytest_defer.txt
for rep in 1 2 3; do
for ds in 0 1; do
  MMTG_DEFER_SUMS=$ds timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --primary-only --no-roofline 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    print('DEFER_SUMS=$ds', d['value'], 'tok/s', d['ms_per_step'], 'ms/step', d['check'].get('launches_per_step'), 'launches/step')
" | tee -a $E/defer_sums_ab.txt
done
done
for ds in 0 1; do MMTG_DEFER_SUMS=$ds timeout 300 python tools/bench_x3.py bf16x3f 64 10 2>&1 | grep -v amdgpu.ids | head -2 | sed "s/^/DEFER_SUMS=$ds /" | tee -a $E/defer_sums_ab.txt; done
;;
s)
# the split-precision backward's column sums batched too: op + model tests, then the bf16x3 step A/B
mkdir -p gpurun_out/r6s2
E=gpurun_out/r6s2
timeout 1500 python -m pytest tests/test_x3_gpu.py tests/test_model_gpu.py tests/test_ddp_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "x3 or batched_column or reproducible or world1 or one_gpu" 2>&1 | tail -4 | tee $E/pytest_defer_x3.txt
for rep in 1 2; do
for ds in 0 1; do MMTG_DEFER_SUMS=$ds timeout 300 python tools/bench_x3.py bf16x3 64 10 2>&1 | grep -v amdgpu.ids | head -1 | sed "s/^/DEFER_SUMS=$ds /" | tee -a $E/defer_sums_x3_ab.txt; done
done
;;
t)
# the last blocks' grouped weight gradients beside the backward's tail (MMTG_WGRAD_TAIL): tests, then the step A/B by number of blocks
mkdir -p gpurun_out/r6t2
E=gpurun_out/r6t2
timeout 1500 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "tail or reproducible or batched_column or fused_step or full_12l" 2>&1 | tail -4 | tee $E/pytest_tail.txt
for rep in 1 2; do
for nt in 0 1 2 3 4; do
  MMTG_WGRAD_TAIL=$nt timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --primary-only --no-roofline 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    print('WGRAD_TAIL=$nt', d['value'], 'tok/s', d['ms_per_step'], 'ms/step', d['check']['probe_myloss_after'])
" | tee -a $E/wgrad_tail_ab.txt
done
done
for nt in 0 2; do MMTG_WGRAD_TAIL=$nt timeout 300 python tools/bench_x3.py bf16x3f 64 10 2>&1 | grep -v amdgpu.ids | head -1 | sed "s/^/WGRAD_TAIL=$nt /" | tee -a $E/wgrad_tail_ab.txt; done
;;
*) echo "usage: $0 {a|c|d|e|f|g|h|i|j|k|l|m|n|o|p|q|r|s|t}"; exit 2 ;;
esac
