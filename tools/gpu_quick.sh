#!/bin/bash
# scratch: in-kernel timelines, persistent kernel
mkdir -p gpurun_out
for args in "15104 3072 768 NT 0 64" "15104 3072 768 NN 1 64" ; do
  timeout 120 python tools/gemm_timeline.py $args 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/gemm_timeline_pp.log
