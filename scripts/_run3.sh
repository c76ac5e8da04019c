export PYTHONUNBUFFERED=1
echo "== new"; python scripts/ab_narrow.py 9 2>/dev/null
bash scripts/abl_stage_counts.sh 164 450 2>&1 | grep -v "^$"
timeout 900 python -m pytest tests/test_gpu_serra09.py -x -q -m gpu 2>&1 | tail -5
timeout 300 python tests/fuzz_serra09.py 120 5 2>&1 | tail -5
