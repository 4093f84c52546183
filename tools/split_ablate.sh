#!/bin/bash
# GPU box helper (round 5): where k_tail16r's time goes, by ablation.  Builds tail_split.o with -DVMC_SPLIT_ABLATE=mask
# for each mask given (bits: 1 no operand split, 2 no gather, 4 no DMA, 8 no barrier, 16 no MFMAs, 32 no LDS reads),
# links each into a library UNDER /tmp (the product library is never touched), and times the split workload's row
# kernel with it through CGS_VMC_DIAGNOSTIC_LIBRARY.  Results of an ablated kernel are garbage: only the time is read.
set -uo pipefail
cd "$(dirname "$0")/.."
C=cgs_vmc_amd/csrc
D=$(mktemp -d /tmp/split_ablate.XXXXXX)
trap 'rm -rf "$D"' EXIT
OBJS=$(ls $C/*.o | grep -v "/tail_split.o")
for m in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DVMC_SPLIT_ABLATE=$m ${ABLATE_EXTRA:-} -c $C/tail_split.hip -o "$D/ts_$m.o" || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS "$D/ts_$m.o" -o "$D/lib_$m.so" || exit 1
  CGS_VMC_DIAGNOSTIC_LIBRARY="$D/lib_$m.so" CGS_VMC_ALLOW_EXTRA_BUILD=1 timeout -k 10 200 \
    python bench.py --workload heisenberg10x10_fc3x256_b4096_split3xbf16 --steps 40 --warmup 5 --reps 1 --no-cpu-baseline --no-extra 2>/dev/null \
    | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ablate mask $m ${ABLATE_EXTRA:-}: tail_eloc %.4f ms' % d['kernels']['tail_eloc']['ms_avg'])"
done
