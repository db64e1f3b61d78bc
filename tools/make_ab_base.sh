#!/bin/bash
# Same-box A/B against an earlier commit: materialise <ref> (default HEAD) with its library built under gpurun_ab/base/
# (git-ignored, shipped to the GPU box), so that one gpurun call can run `python gpurun_ab/base/bench.py ...` beside the
# working tree's `python bench.py ...`.
set -e
ref=${1:-HEAD}
rm -rf gpurun_ab/base; mkdir -p gpurun_ab/base
git archive $ref mmtg_amd include bench.py oracle tools/gpu_ab.sh tests/helpers.py | tar -x -C gpurun_ab/base
(cd gpurun_ab/base && python -m mmtg_amd.build --jobs 8 2>&1 | tail -1)
git rev-parse $ref > gpurun_ab/base/REF
