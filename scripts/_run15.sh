export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_serra09.py tests/test_serra09_substeps.py tests/test_gpu_configs.py -x -q -m gpu -k "not earlyfusion and not simple" 2>&1 | tail -5
timeout 400 python tests/fuzz_serra09.py 150 11 2>&1 | tail -3
for v in 1 2; do echo "== ACX_BAND2=$v"; ACX_BAND2=$v python scripts/ab_narrow.py 9 2>/dev/null; ACX_BAND2=$v python scripts/quick_bench_rand.py 120 600 2>/dev/null | grep "pairs/s" | tail -1; ACX_BAND2=$v python scripts/quick_bench_rand.py 100 750 2>/dev/null | grep "pairs/s" | tail -1; done
echo "== b2w7"; ACX_LIB=build_ab/libacx_b2w7.so python scripts/ab_narrow.py 9 2>/dev/null
