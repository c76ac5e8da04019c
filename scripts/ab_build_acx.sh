#!/bin/bash
# Development aid: A/B builds of acx.hip (everything but the band kernel) linked against the current
# acx_band.o -> build_ab/libacx_<name>.so; pick one at run time with ACX_LIB=build_ab/libacx_<name>.so.
#   scripts/ab_build_acx.sh name [extra hipcc flags ...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p "$ROOT/build_ab"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -w"
/opt/rocm/bin/hipcc $FL "$@" -c -o "/tmp/acx_$name.o" "$ROOT/acoss_amd/csrc/acx.hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build_ab/libacx_$name.so" "/tmp/acx_$name.o" "$ROOT/acoss_amd/csrc/acx_band.o"
echo "built build_ab/libacx_$name.so"
