export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_grid.py -x -q -m gpu 2>&1 | tail -8
ACX_BENCH_FORCE_COLLECTIVE=1 python bench.py --no-cpu --no-other > gpurun_out/r05c_bench_nccl1.json 2> gpurun_out/r05c_bench_nccl1.err; echo "rc=$?"
python - <<'PY'
import json
l=json.load(open("gpurun_out/r05c_bench_nccl1.json"))
print(l["value"], l["collectives"], l["rccl_version"], l["ranks"], json.dumps(l["strong"], indent=1))
PY
