export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05b_covers -o s -- python3 scripts/quick_bench_covers.py 82 150 650 > gpurun_out/r05b_covers.log 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r05b_covers/s_kernel_stats.csv")))
for r in rows[:14]:
    print("%-70s calls %4s  total %8.3f ms  avg %8.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
