export PYTHONUNBUFFERED=1 TMPDIR=/tmp
for rep in 1 2; do for v in k1i1 k0i1 k1i0; do ACX_LIB=build_ab/libacx_$v.so python scripts/ef_gemm_probe.py 128 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['pairs_per_s'], d['kernels_ms'])"; done; done
ACX_EF_ROWSTAT2=0 ACX_LIB=build_ab/libacx_k1i1.so python scripts/ef_gemm_probe.py 128 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('off', d['pairs_per_s'], d['kernels_ms'])"
export ACX_LIB=build_ab/libacx_k1i1.so
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/r05e_ef2pmc -o p -- python3 scripts/ef_gemm_probe.py 128 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d gpurun_out/r05e_ef2pmc2 -o p -- python3 scripts/ef_gemm_probe.py 128 1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/r05e_ef2pmc","gpurun_out/r05e_ef2pmc2"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("void ","")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        if "rowstat" in k or "colstat" in k:
            print(k, {c: "%.4g" % (sum(x)/len(x)) for c,x in v.items()})
PY
