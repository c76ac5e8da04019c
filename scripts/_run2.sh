export PYTHONUNBUFFERED=1
echo "== w6"; ACX_LIB=build_ab/libacx_abl0.so python scripts/ab_narrow.py 9 2>/dev/null
echo "== w8"; ACX_LIB=build_ab/libacx_b2w8.so python scripts/ab_narrow.py 9 2>/dev/null
echo "== old"; ACX_BAND2=0 ACX_LIB=build_ab/libacx_b2w8.so python scripts/ab_narrow.py 9 2>/dev/null
bash scripts/abl_stage_counts.sh 164 450 2>&1 | grep -v "^$"
ACX_LIB=build_ab/libacx_abl0.so bash scripts/pmc.sh b2w6_450 scripts/quick_bench_rand.py 164 450 2>&1 | tail -80
