#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
C=cgs_vmc_amd/csrc
D=$(mktemp -d /tmp/ssd.XXXXXX)
trap 'rm -rf "$D"' EXIT
OBJS=$(ls $C/*.o | grep -v "/sweep_split.o")
for rt in 8 0; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSWEEP_SPLIT_RT=$rt -c $C/sweep_split.hip -o "$D/ss_$rt.o" 2>/dev/null || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS "$D/ss_$rt.o" -o "$D/lib_$rt.so" || exit 1
  echo "== SWEEP_SPLIT_RT=$rt"
  CGS_VMC_DIAGNOSTIC_LIBRARY="$D/lib_$rt.so" CGS_VMC_ALLOW_EXTRA_BUILD=1 timeout -k 10 120 python tools/split_sampler_debug.py 2>&1 | grep -v "amdgpu.ids\|DIAGNOSTIC"
done
