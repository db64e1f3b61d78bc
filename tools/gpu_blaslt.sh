#!/bin/bash
mkdir -p gpurun_out/blaslt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/micro/blaslt_calib.py 2>&1 | tee $R/gpurun_out/blaslt/times.txt
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/blaslt/prof -o cal --output-format csv -- python3 $R/tools/micro/blaslt_calib.py > /dev/null 2>&1
cp $R/gpurun_out/blaslt/prof/*kernel_stats.csv $R/gpurun_out/blaslt/kernel_stats.csv 2>/dev/null
find $R/gpurun_out/blaslt/prof -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/blaslt/kernel_stats.csv \;
rm -rf $R/gpurun_out/blaslt/prof
