#!/bin/bash
mkdir -p gpurun_out/r3l
export TMPDIR=/tmp
R=$(pwd)
for t in base new; do
  if [ $t = base ]; then d=$R/gpurun_ab/base; else d=$R; fi
  cd $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3l/$t -o dec -- python3 bench.py --mode decode --steps 2 --warmup 1 --no-roofline --no-cpu-baseline > $R/gpurun_out/r3l/$t.log 2>&1
  cd $R
  f=$(find gpurun_out/r3l/$t -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r3l/${t}_kernel_stats.csv
  find gpurun_out/r3l/$t -name "*kernel_trace.csv" -delete
  head -9 gpurun_out/r3l/${t}_kernel_stats.csv | cut -c1-160
done
