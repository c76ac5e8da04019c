export PYTHONUNBUFFERED=1 TMPDIR=/tmp
python scripts/ab_narrow.py 11 2>/dev/null
rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d gpurun_out/r05b_hip -o s -- python3 scripts/quick_bench_covers.py 82 150 650 > gpurun_out/r05b_hip.log 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r05b_hip/s_hip_api_stats.csv")))
for r in rows[:16]:
    print("%-40s calls %5s  total %9.3f ms  avg %8.1f us" % (r["Name"][:40], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
