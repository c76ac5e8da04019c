export PYTHONUNBUFFERED=1 TMPDIR=/tmp
python scripts/end_to_end.py 15000 covers > gpurun_out/r05_e2e_covers.log 2>&1; tail -1 gpurun_out/r05_e2e_covers.log | cut -c1-400
cp gpurun_out/end_to_end.json gpurun_out/r05_end_to_end_15000_covers.json
bash scripts/profile.sh r05 > gpurun_out/r05_profile.log 2>&1
python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err; echo "bench rc=$?"
bash scripts/pmc.sh r05_b2_450 scripts/quick_bench_rand.py 164 450 > gpurun_out/r05_pmc_b2_450.txt 2>&1
bash scripts/pmc.sh r05_b2_covers scripts/quick_bench_covers.py 82 150 650 > gpurun_out/r05_pmc_b2_covers.txt 2>&1
