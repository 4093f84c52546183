#!/bin/bash
# Builds variant libraries that differ from the product library in sweep8.o only (ring depth, resident fragments of
# layer 1), for A/B timing through CGS_VMC_DIAGNOSTIC_LIBRARY (never the product path):
#   tools/sweep8_variants.sh "<tag> <flags>" ...     e.g. "pf4 -DSWEEP8_PF=4" "r1_8 -DSWEEP8_R1_256=8"
# Output: build_variants/lib_<tag>.so (gitignored; travels to the GPU box).
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/cgs_vmc_amd/csrc
mkdir -p "$ROOT/build_variants"
OBJS=$(ls $CS/*.o | grep -v "/sweep8.o")
for spec in "$@"; do
  set -- $spec
  tag=$1; shift
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value "$@" -c $CS/sweep8.hip -o $ROOT/build_variants/sweep8_$tag.o \
    -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "ILi8ELi32.*Lb0ELb0E|VGPRs Spill|ScratchSize" | head -40 | paste - - - | grep "ILi8" || true
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $ROOT/build_variants/sweep8_$tag.o -o $ROOT/build_variants/lib_$tag.so
  rm -f $ROOT/build_variants/sweep8_$tag.o
  echo "built lib_$tag.so"
done
