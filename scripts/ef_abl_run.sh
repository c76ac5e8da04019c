#!/bin/bash
# GPU box: kernel-trace stats of scripts/ef_gemm_probe.py for every build_ab/libacx_<name>.so given (and the product build)
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
PY=$(which python3)
OUT=$PWD/gpurun_out/ef_abl
mkdir -p $OUT
for name in base "$@"; do
  if [ $name = base ]; then unset ACX_LIB; else export ACX_LIB=$PWD/build_ab/libacx_$name.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o s -- $PY scripts/ef_gemm_probe.py ${N:-48} 3 > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name "*kernel_stats.csv" | head -1)
  echo "== $name"; grep -E "ef_gemm|ef_rowstat|sw_kernel|ef_fuse" $f | awk -F, '{print $1, $2, $4}' | cut -c1-60,200-
  grep -E '^\{' $OUT/$name.log | cut -c1-120
done
