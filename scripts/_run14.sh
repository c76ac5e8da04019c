export PYTHONUNBUFFERED=1 TMPDIR=/tmp
python scripts/end_to_end.py 15000 covers > gpurun_out/r05_e2e_covers.log 2>&1; tail -3 gpurun_out/r05_e2e_covers.log
ls gpurun_out/*end_to_end* 2>/dev/null
bash scripts/profile.sh r05 > gpurun_out/r05_profile.log 2>&1; tail -3 gpurun_out/r05_profile.log
