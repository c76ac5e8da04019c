#!/bin/bash
# Per-stage instruction counts of band_kernel from the ablation builds (scripts/ab_build.sh ablN -DACX_ABL=N):
#   scripts/abl_stage_counts.sh <n_tracks> <T>       (on the GPU box; prints one line per build and kernel)
set -u
N=${1:-164}; T=${2:-450}
export TMPDIR=/tmp
PY=$(which python3)
for n in 1 2 3 4 0; do
  OUT=$PWD/gpurun_out/abl_${T}_$n
  mkdir -p $OUT
  ACX_LIB=$PWD/build_ab/libacx_abl$n.so rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT -o pmc -- $PY scripts/quick_bench_rand.py $N $T > $OUT/log.txt 2>&1
  ACX_LIB=$PWD/build_ab/libacx_abl$n.so $PY scripts/quick_bench_rand.py $N $T 2>/dev/null | grep band | tail -1 | sed "s/^/stage $n time: /"
  $PY - <<PYEOF
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for p in glob.glob("$OUT/*counter_collection.csv") + glob.glob("$OUT/*/*counter_collection.csv"):
    for row in csv.DictReader(open(p)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("acx::", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in sorted(acc.items()):
    if not k.startswith("band"): continue
    m = {n: sum(v) / len(v) for n, v in d.items()}
    w = m["SQ_WAVES"]
    print("stage $n %-30s VALU/wave %6.1f SALU/wave %6.1f LDS/wave %5.1f MFMA/wave %5.1f VMEM/wave %4.1f valu-quad-cycles/wave %6.1f wave-cycles/wave %7.1f"
          % (k, m["SQ_INSTS_VALU"] / w, m["SQ_INSTS_SALU"] / w, m["SQ_INSTS_LDS"] / w, m["SQ_INSTS_MFMA"] / w, m["SQ_INSTS_VMEM_RD"] / w, m["SQ_ACTIVE_INST_VALU"] / w, m["SQ_WAVE_CYCLES"] / w))
PYEOF
done
