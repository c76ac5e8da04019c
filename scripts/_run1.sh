set -x
mkdir -p gpurun_out/r05a
export PYTHONUNBUFFERED=1
for v in 0 1; do
  echo "== ACX_BAND2=$v"
  ACX_BAND2=$v python scripts/quick_bench_rand.py 164 450 2>&1 | grep -v "^   [a-z]" 
  ACX_BAND2=$v python scripts/quick_bench_rand.py 200 250 2>&1 | grep -v "^   [a-z]"
  ACX_BAND2=$v python scripts/quick_bench_covers.py 82 150 650 2>&1 | grep -E "crc|pairs/s"
done
echo "== b2w6"
ACX_LIB=build_ab/libacx_b2w6.so python scripts/quick_bench_rand.py 164 450 2>&1 | grep -v "^   [a-z]"
ACX_LIB=build_ab/libacx_b2w6.so python scripts/quick_bench_rand.py 200 250 2>&1 | grep -v "^   [a-z]"
ACX_LIB=build_ab/libacx_b2w6.so python scripts/quick_bench_covers.py 82 150 650 2>&1 | grep -E "crc|pairs/s"
timeout 900 python -m pytest tests/test_gpu_serra09.py tests/test_serra09_substeps.py -x -q -m gpu 2>&1 | tail -15
