#!/bin/bash
# Per-stage instruction counts of ef_rowstat_kernel from the ablation builds (scripts/ab_build_acx.sh ablN -DACX_EF_ABL=N, N = 1 .. 4):
#   scripts/abl_stage_counts_ef.sh [n_tracks]        (on the GPU box; one line per build and instantiation)
set -u
N=${1:-64}
export TMPDIR=/tmp
PY=$(which python3)
for n in 1 2 3 4 0; do
  OUT=$PWD/gpurun_out/ablef_$n
  mkdir -p $OUT
  L=$PWD/build_ab/libacx_abl$n.so; [ $n = 0 ] && L=$PWD/acoss_amd/csrc/libacx.so
  ACX_LIB=$L rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT -o pmc -- $PY scripts/ef_gemm_probe.py $N 1 > $OUT/log.txt 2>&1
  $PY - <<PYEOF
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for p in glob.glob("$OUT/*counter_collection.csv") + glob.glob("$OUT/*/*counter_collection.csv"):
    for row in csv.DictReader(open(p)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("acx::", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in sorted(acc.items()):
    if not k.startswith("ef_rowstat_kernel"): continue
    m = {n: sum(v) / len(v) for n, v in d.items()}
    w = m["SQ_WAVES"]
    print("stage $n %-36s VALU/row %6.1f SALU/row %6.1f LDS/row %5.1f VMEM rd/row %4.1f wr/row %4.1f"
          % (k, m["SQ_INSTS_VALU"] / w, m["SQ_INSTS_SALU"] / w, m["SQ_INSTS_LDS"] / w, m["SQ_INSTS_VMEM_RD"] / w, m["SQ_INSTS_VMEM_WR"] / w))
PYEOF
done
