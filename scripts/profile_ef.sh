#!/bin/bash
# GPU box: kernel stats + PMC passes (each in its own run, counters never together with a trace) of
# scripts/ef_gemm_probe.py N 2 (default) or of the python program given behind the tag ("other": bench_other.py);
# per-kernel averages of every acx kernel -> gpurun_out/prof_ef_<tag>/summary.txt
#   scripts/profile_ef.sh TAG [N | other]
set -u
TAG=${1:-x}
N=${2:-48}
OUT=$PWD/gpurun_out/prof_ef_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
PY=$(which python3)
ARGS="scripts/ef_gemm_probe.py $N 2"
if [ "$N" = other ]; then ARGS="bench_other.py --steps 2 --warmup 1"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $PY $ARGS > $OUT/stats.log 2>&1
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -o p -- $PY $ARGS > $OUT/pmc$i.log 2>&1
done
$PY - $OUT <<'PYEOF' | tee $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
st = {}
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        st[r["Name"].split("(")[0].replace("void ", "")] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
for k in sorted(agg):
    if "acx::" not in k and "anonymous" not in k: continue
    print("==", k, "calls %d avg_ms %.3f" % st.get(k, (0, 0)))
    for c in sorted(agg[k]):
        v = agg[k][c]
        print("   %-28s %.4g" % (c, sum(v) / len(v)))
PYEOF
