#!/bin/bash
# Long runs of the three randomised GPU-vs-oracle checkers on the build in the tree (a development aid; on the GPU box):
#   bash scripts/fuzz_soak.sh [seconds per checker] [seed]   -> gpurun_out/fuzz_soak.txt
S=${1:-600}; SEED=${2:-606}
export PYTHONUNBUFFERED=1
out=gpurun_out/fuzz_soak.txt; mkdir -p gpurun_out; : > $out
echo "== tests/fuzz_serra09.py $S $SEED" | tee -a $out
timeout $((S + 300)) python tests/fuzz_serra09.py $S $SEED 2>&1 | tail -4 | tee -a $out
echo "== tests/fuzz_other.py $S $SEED" | tee -a $out
timeout $((S + 300)) python tests/fuzz_other.py $S $SEED 2>&1 | tail -6 | tee -a $out
R=$((S / 6))
echo "== tests/fuzz_earlyfusion.py $R $SEED" | tee -a $out
timeout $((S * 2 + 300)) python tests/fuzz_earlyfusion.py $R $SEED 2>&1 | tail -4 | tee -a $out
