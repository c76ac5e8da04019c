export PYTHONUNBUFFERED=1 TMPDIR=/tmp
echo "== wide class: mm vs nomm"
for v in nomm mm nomm mm; do ACX_LIB=build_ab/libacx_$v.so python scripts/quick_bench_rand.py 64 2000 2>/dev/null | grep "pairs/s" | tail -1 | sed "s/^/$v /"; done
for v in nomm mm; do ACX_LIB=build_ab/libacx_$v.so python scripts/quick_bench_rand.py 120 900 2>/dev/null | grep "pairs/s" | tail -1 | sed "s/^/$v T=900 /"; done
for v in nomm mm; do echo "== $v"; ACX_LIB=build_ab/libacx_$v.so python scripts/ab_narrow.py 7 2>/dev/null; done
echo "== EF probe rowstat2 on/off"
ACX_EF_ROWSTAT2=0 python scripts/ef_gemm_probe.py 128 3 2>/dev/null | tail -12
ACX_EF_ROWSTAT2=1 python scripts/ef_gemm_probe.py 128 3 2>/dev/null | tail -12
timeout 1200 python -m pytest tests/test_gpu_earlyfusion.py tests/test_gpu_serra09.py -x -q -m gpu 2>&1 | tail -5
