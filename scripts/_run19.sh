export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -o s -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/prof_final_bench.json 2> gpurun_out/prof_final_bench.err; echo "rc=$?"
python scripts/timed_region_stats.py gpurun_out/prof_final/s_kernel_trace.csv > gpurun_out/r05_bench_kernel_stats.md; cat gpurun_out/r05_bench_kernel_stats.md
