#!/bin/bash
# Development aid: A/B builds of the band kernel's translation unit (m = 9 only) linked against the
# current acx.o -> build_ab/libacx_<name>.so; pick one at run time with ACX_LIB=build_ab/libacx_<name>.so.
#   scripts/ab_build.sh name [extra hipcc flags ...]        (SRC=dir overrides the source directory)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
SRC=${SRC:-$ROOT/acoss_amd/csrc}
mkdir -p "$ROOT/build_ab"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -DACX_FAST_BUILD -mllvm -amdgpu-sched-strategy=max-ilp -w"
/opt/rocm/bin/hipcc $FL "$@" -I"$SRC" -I"$ROOT/acoss_amd/csrc" -c -o "/tmp/band_$name.o" "$SRC/acx_band.hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build_ab/libacx_$name.so" "$ROOT/acoss_amd/csrc/acx.o" "/tmp/band_$name.o"
echo "built build_ab/libacx_$name.so"
