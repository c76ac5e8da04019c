export PYTHONUNBUFFERED=1 TMPDIR=/tmp
bash scripts/profile_ef.sh r05 128 > gpurun_out/r05_profile_ef.log 2>&1
bash scripts/pmc.sh r05_b2_450 scripts/quick_bench_rand.py 164 450 > gpurun_out/r05_pmc_b2_450.txt 2>&1
bash scripts/pmc.sh r05_b2_covers scripts/quick_bench_covers.py 82 150 650 > gpurun_out/r05_pmc_b2_covers.txt 2>&1
bash scripts/pmc.sh r05_wide scripts/quick_bench_rand.py 64 2000 > gpurun_out/r05_pmc_wide.txt 2>&1
tail -5 gpurun_out/prof_ef_r05/summary.txt
