#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for bench.py.
# Outputs under gpurun_out/prof_<tag>/ ; summarise with scripts/summarise_profile.py.
# (Counters are collected in their own runs, never together with a trace; the program itself follows `--`.)
set -u
TAG=${1:-r03}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
PY=$(which python3)
ARGS="bench.py --no-cpu --no-other --steps 2 --warmup 1 --tracks 640"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $PY $ARGS > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $PY $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $PY $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq1 -o pmc -- $PY $ARGS > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -o pmc -- $PY $ARGS > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq3 -o pmc -- $PY $ARGS > $OUT/pmc_sq3.log 2>&1
$PY $ARGS > $OUT/bench.json 2> $OUT/bench.err
# the other two algorithms: kernel-trace stats of bench_other.py
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_other -o stats -- $PY bench_other.py --steps 2 --warmup 1 > $OUT/stats_other.log 2>&1
find $OUT -name "*.csv" | head -30
du -sh $OUT
