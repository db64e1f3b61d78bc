#!/bin/bash
# run a selection of GPU tests: bash tools/gpu_tests.sh "<files>" "<-k expression>" [tail lines]
mkdir -p gpurun_out/t
export MMTG_TEST_REPORT=$(pwd)/gpurun_out/t/test_report.jsonl
rm -f $MMTG_TEST_REPORT
if [ -n "$2" ]; then
timeout 2400 python -m pytest $1 -k "$2" -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -${3:-60} | tee gpurun_out/t/pytest.txt
else
timeout 2400 python -m pytest $1 -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -${3:-60} | tee gpurun_out/t/pytest.txt
fi
cat $MMTG_TEST_REPORT 2>/dev/null
