#!/bin/bash
# usage: scripts/pmc_one.sh <tag> "<counters>" <python args...>  -- one rocprofv3 --pmc pass, per-kernel means
set -u
TAG=$1; CNT=$2; shift 2
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
PY=$(which python3)
rocprofv3 --pmc $CNT --output-format csv -d $OUT/g -o pmc -- $PY "$@" > $OUT/g.log 2>&1
$PY - <<PYEOF
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for p in glob.glob("$OUT/g/*counter_collection.csv") + glob.glob("$OUT/g/*/*counter_collection.csv"):
    for row in csv.DictReader(open(p)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("acx::", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print("==", k)
    for n in sorted(d):
        v = d[n]; print("   %-32s %.4g  (n=%d)" % (n, sum(v) / len(v), len(v)))
PYEOF
