export PYTHONUNBUFFERED=1 TMPDIR=/tmp
for r in 0 1 0 1; do ACX_EF_ROWSTAT2=$r python scripts/ef_gemm_probe.py 128 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['env'], d['pairs_per_s'], d['kernels_ms'])"; done
python scripts/quick_bench_rand.py 64 2000 2>/dev/null | grep "pairs/s" | tail -1
timeout 2400 python -m pytest tests/test_gpu_earlyfusion.py tests/test_gpu_parity_sets.py tests/test_gpu_configs.py -x -q -m gpu -k "earlyfusion or ef" 2>&1 | tail -6
timeout 600 python tests/fuzz_earlyfusion.py 90 3 2>&1 | tail -3
