export PYTHONUNBUFFERED=1 TMPDIR=/tmp
bash scripts/profile_ef.sh r05o other > gpurun_out/r05o_profile.log 2>&1; tail -2 gpurun_out/r05o_profile.log
python bench.py > gpurun_out/r05_bench2.json 2> gpurun_out/r05_bench2.err; echo "bench rc=$?"
ACX_BENCH_FORCE_COLLECTIVE=1 python bench.py --no-cpu --no-other > gpurun_out/r05_bench_nccl_world1.json 2>/dev/null; echo "rc=$?"
timeout 600 python -m pytest tests/test_gpu_serra09.py -x -q -m gpu -k "f16x2" 2>&1 | tail -3
