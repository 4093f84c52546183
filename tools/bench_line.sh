#!/bin/bash
# One short bench line per workload: tools/bench_line.sh <workload> [steps] -> "workload ms_per_step {kernel ms}"
wl=${1:?workload}; steps=${2:-5}
timeout -k 10 280 python3 bench.py --workload "$wl" --steps "$steps" --warmup 1 --warm-sweeps 1 --reps 3 --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['config']['workload'], '%.2f ms per step;' % d['ms_per_step'], ', '.join('%s %.2f' % (n, v['ms_avg']) for n, v in d['kernels'].items()), '; rows', d['connected_rows_last_eloc'], '; energy per site', d['mean_energy_per_site'])
"
