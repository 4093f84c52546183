#!/bin/bash
# GPU box helper (round 5): k_sweep16s with other numbers of resident k-steps (SWEEP_SPLIT_RT = 2 x resident 32-deep
# k-steps of layer 0), each built into a library under /tmp and timed on the split-sampler workload.
set -uo pipefail
cd "$(dirname "$0")/.."
C=cgs_vmc_amd/csrc
D=$(mktemp -d /tmp/ssv.XXXXXX)
trap 'rm -rf "$D"' EXIT
OBJS=$(ls $C/*.o | grep -v "/sweep_split.o")
for rt in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSWEEP_SPLIT_RT=$rt ${SSV_EXTRA:-} -c $C/sweep_split.hip -o "$D/ss_$rt.o" 2>/dev/null || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS "$D/ss_$rt.o" -o "$D/lib_$rt.so" || exit 1
  CGS_VMC_DIAGNOSTIC_LIBRARY="$D/lib_$rt.so" CGS_VMC_ALLOW_EXTRA_BUILD=1 timeout -k 10 200 \
    python bench.py --workload heisenberg10x10_fc3x256_b4096_split3xbf16_sampler --steps 60 --warmup 5 --reps 3 --no-cpu-baseline --no-extra 2>/dev/null \
    | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SWEEP_SPLIT_RT=$rt ${SSV_EXTRA:-}: sweep %.4f ms  step %.4f ms  E/N %.5f' % (d['kernels']['sweep']['ms_avg'], d['ms_per_step'], d['mean_energy_per_site']))"
done
