export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 3000 python -m pytest tests/test_gpu_parity_sets.py tests/test_gpu_configs.py -x -q -m gpu -k "earlyfusion" 2>&1 | tail -15
ACX_EF_PARITY_LARGE=1 timeout 3400 python -m pytest tests/test_gpu_parity_sets.py -x -q -m gpu -k "cover_set_map_1500" 2>&1 | tail -15
cp gpurun_out/parity_ef.json gpurun_out/r05_parity_ef.json
