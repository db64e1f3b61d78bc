#!/bin/bash
# round 4, probe 1: where the in-situ (HBM-cold) time of the eight-phase products goes -- timelines warm vs cold, per-shape table warm vs cold
mkdir -p gpurun_out/p1
O=gpurun_out/p1
( NTSET=1 python tools/bench_gemm.py; echo "--- COLD"; COLD=1 NTSET=1 python tools/bench_gemm.py ) > $O/ntset_warm_cold.txt 2>&1
for spec in "15104 768 768 NT 3" "15104 768 768 NT 7" "15104 768 3072 NT 3" "15104 2304 768 NT 0" "15104 3072 768 NT 1" "15104 3072 768 NT 4" "15104 768 2304 NT 0"; do
  echo "=== warm $spec"; python tools/gemm_timeline.py $spec 2>&1 | grep -v amdgpu.ids
  echo "=== COLD $spec"; COLD=1 python tools/gemm_timeline.py $spec 2>&1 | grep -v amdgpu.ids
done > $O/timelines_warm_cold.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
tail -c 3000 $O/bench_default.json
