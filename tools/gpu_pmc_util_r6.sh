#!/bin/bash
# MFMA / LDS utilisation, L2 hit rate and HBM bytes of the bf16 step's dominant kernels (tools/pmc_case_r6.py): one rocprofv3 --pmc
# pass per counter group (only --kernel-trace beside it, the program directly after `--`), every pass under a timeout; last dispatch
# of each kernel reported.
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/pmcu6; rm -rf $O; mkdir -p $O
i=0
while read -r set; do
  i=$((i+1))
  ( cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O -o s$i -- python3 $R/tools/pmc_case_r6.py > $O/log$i.txt 2>&1 ) || echo "set $i failed: $set"
done <<'SETS'
MfmaUtil LdsUtil
SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS
GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA SQ_INSTS_VALU
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
LdsLatency MemUnitStalled
FETCH_SIZE
WRITE_SIZE
SETS
python3 - <<'PY'
import csv, glob, collections, re
res = collections.OrderedDict()
def short(k):
    k = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+|void ", "", k)
    return k.split("(")[0][:46]
for f in sorted(glob.glob('gpurun_out/pmcu6/*counter_collection.csv')):
    last = {}
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if not any(s in k for s in ("gemm_p8", "gemm_occ4", "wgrad_group", "attn_fwd_small", "attn_bwd_small", "ln_bwd3", "ln_fwd3")): continue
        last[(short(k), r['Counter_Name'])] = float(r['Counter_Value'])
    for (s, c), v in last.items():
        res.setdefault(c, {})[s] = v
kern = sorted({k for d in res.values() for k in d})
print("kernels of the bf16 step (last dispatch of each; tools/pmc_case_r6.py lists the products in launch order):")
for n, k in enumerate(kern): print("  [%d] %s" % (n, k))
print("%-34s" % "counter", *["%12s" % ("[%d]" % n) for n in range(len(kern))])
for c, d in res.items():
    print("%-34s" % c, *["%12.4g" % d.get(k, float('nan')) for k in kern])
PY
