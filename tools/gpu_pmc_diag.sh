#!/bin/bash
# Where the GEMM kernels' cycles go: one rocprofv3 --pmc pass per counter group (no trace domains besides
# --kernel-trace), last dispatch of each kernel reported.
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/pmcd; mkdir -p $O
i=0
while read -r set; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O -o s$i -- python3 tools/pmc_case.py > $O/log$i.txt 2>&1 || echo "set $i failed: $set"
done <<'SETS'
SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS
SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
TA_BUSY_avr TCC_BUSY_avr TCP_TCR_TCP_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT TCP_TCP_TA_DATA_STALL_CYCLES_sum
MfmaUtil LdsUtil
LdsLatency MemUnitStalled
TA_BUFFER_READ_LDS_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_TA_BUSY_sum TD_TD_BUSY_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA SQ_INSTS_VALU
SETS
python3 - <<'PY'
import csv, glob, collections
res = collections.OrderedDict()
for f in sorted(glob.glob('gpurun_out/pmcd/*counter_collection.csv')):
    last = {}
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'gemm' not in k: continue
        short = k.split('gemm_')[1][:44]
        last[(short, r['Counter_Name'])] = (int(r['Dispatch_Id']), float(r['Counter_Value']))
    for (short, c), (d, v) in last.items():
        res.setdefault(c, {})[short] = v
kern = sorted({k for d in res.values() for k in d})
print("%-40s" % "counter", *["%22s" % k[:22] for k in kern])
for c, d in res.items():
    print("%-40s" % c, *["%22.4g" % d.get(k, float('nan')) for k in kern])
PY
