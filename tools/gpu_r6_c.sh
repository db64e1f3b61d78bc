#!/bin/bash
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
