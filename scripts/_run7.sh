export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
python bench.py > gpurun_out/r05c_bench.json 2> gpurun_out/r05c_bench.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/r05c_bench.err
python - <<'PY'
import json
l=json.load(open("gpurun_out/r05c_bench.json"))
print(l["value"], l["ms_per_step"], l["roofline"]["frac"], l["ranks"], l["strong"])
for k,v in l["other"].items():
    print(k, {a:b for a,b in v.items() if a in ("value","unit","gcells_per_s","error")}, v.get("roofline",{}).get("frac"))
print(l["phases_s"])
PY
