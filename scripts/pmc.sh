#!/bin/bash
# usage: scripts/pmc.sh <tag> <python args...>   -- PMC passes (SQ groups) for one command, on the GPU box
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
PY=$(which python3)
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -o pmc -- $PY "$@" > $OUT/g$i.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $PY "$@" > $OUT/stats.log 2>&1
$PY - <<PYEOF
import csv, os, glob
from collections import defaultdict
out = "$OUT"
acc = defaultdict(lambda: defaultdict(list))
for p in glob.glob(out + "/g*/pmc_counter_collection.csv"):
    for row in csv.DictReader(open(p)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("acx::", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        acc[k]["VGPR"] = [float(row["VGPR_Count"]) + float(row["Accum_VGPR_Count"])]
for row in csv.DictReader(open(out + "/stats/stats_kernel_stats.csv")):
    k = row["Name"].split("(")[0].replace("void ", "").replace("acx::", "")
    acc[k]["avg_ms"] = [float(row["AverageNs"]) / 1e6]
for k, d in acc.items():
    if "kernel" not in k: continue
    print("==", k)
    for n in sorted(d):
        v = d[n]; print("   %-32s %.4g" % (n, sum(v) / len(v)))
PYEOF
