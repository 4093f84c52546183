#!/bin/bash
# The general convolution sampler's chain groups (run_sweep_cgen, CGS_VMC_CONV_GENERAL_GROUPS): step and sweep time of
# a bench workload for G = 1 .. 4.  Usage (on the GPU box): tools/conv_groups.sh <workload> [steps] [groups ...]
set -e
wl=${1:?workload}; steps=${2:-5}; shift; shift || true
groups=${*:-1 2 3 4}
for g in $groups; do
  CGS_VMC_CONV_GENERAL_GROUPS=$g timeout -k 10 280 python3 bench.py --workload "$wl" --steps "$steps" --warmup 1 \
    --warm-sweeps 1 --reps 3 --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('$wl G=$g: %.2f ms per step; sweep %.2f ms, local energies %.2f ms' % (d['ms_per_step'], k['sweep']['ms_avg'], k['tail_eloc']['ms_avg']))
"
done
