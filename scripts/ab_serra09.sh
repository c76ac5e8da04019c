#!/bin/bash
# Development aid: A/B of libacx builds under build_ab/ on four Serra09 workloads.  usage: scripts/ab_serra09.sh base variant ...
for v in "$@"; do
  echo "== $v"
  ACX_LIB=build_ab/$v.so python scripts/quick_bench_covers.py 2>&1 | tail -3 | grep band | sed 's/^/  covers80-shaped   /'
  ACX_LIB=build_ab/$v.so python scripts/quick_bench_covers.py 40 800 1000 2>&1 | tail -3 | grep band | sed 's/^/  covers 800-1000   /'
  ACX_LIB=build_ab/$v.so python scripts/quick_bench_rand.py 164 450 2>&1 | tail -3 | grep band | sed 's/^/  rand 450          /'
  ACX_LIB=build_ab/$v.so python bench.py --steps 3 --warmup 1 2>&1 | tail -1 | cut -c1-110
done
