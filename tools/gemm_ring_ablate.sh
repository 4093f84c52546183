#!/bin/bash
# GPU box helper (round 5): where k_gemm_ring's time goes, by ablation.  Builds grad.o with -DVMC_GR_ABLATE=mask for
# each mask given (bits: 1 no DMA in the stage loop, 2 no barrier, 4 no operand reads, 8 no MFMAs), links each into a
# library UNDER /tmp (the product library is never touched), and times the general sampler with it through
# CGS_VMC_DIAGNOSTIC_LIBRARY.  Results of an ablated kernel are garbage: only the time is read.
set -uo pipefail
cd "$(dirname "$0")/.."
C=cgs_vmc_amd/csrc
D=$(mktemp -d /tmp/gr_ablate.XXXXXX)
trap 'rm -rf "$D"' EXIT
OBJS=$(ls $C/*.o | grep -v "/grad.o")
for m in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DVMC_GR_ABLATE=$m ${ABLATE_EXTRA:-} -c $C/grad.hip -o "$D/g_$m.o" 2>/dev/null || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS "$D/g_$m.o" -o "$D/lib_$m.so" || exit 1
  CGS_VMC_DIAGNOSTIC_LIBRARY="$D/lib_$m.so" CGS_VMC_ALLOW_EXTRA_BUILD=1 timeout -k 10 200 \
    python bench.py --workload heisenberg10x10_fc3x1024_b4096 --steps 20 --warmup 3 --reps 1 --no-cpu-baseline --no-extra 2>/dev/null \
    | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ablate mask $m ${ABLATE_EXTRA:-}: sweep %.3f ms = %.1f us per mc_step' % (d['kernels']['sweep']['ms_avg'], d['kernels']['sweep']['ms_avg'] * 10))"
done
