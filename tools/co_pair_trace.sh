#!/bin/bash
# usage: co_pair_trace.sh tag  -- bench + kernel trace durations of the pair
set -euo pipefail
tag=$1
mkdir -p gpurun_out/co_pair_$tag
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}"
CGS_VMC_CO=1 timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/co_pair_$tag/prof -o tr --output-format csv -- python bench.py --no-cpu-baseline --no-timing --steps 10 --warmup 3 --reps 1 > gpurun_out/co_pair_$tag/bench.json 2> gpurun_out/co_pair_$tag/bench.err
f=$(find gpurun_out/co_pair_$tag/prof -name '*kernel_trace.csv' | head -1)
test -n "$f"
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    d[r['Kernel_Name'][:36]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in d.items():
    if 'sweep' in k or 'tail' in k:
        v2 = sorted(v)
        print('%-38s n=%3d median %.1f us  min %.1f max %.1f' % (k, len(v), v2[len(v2)//2], v2[0], v2[-1]))
PY
python -c "import json; print('ms_per_step', json.loads(open('gpurun_out/co_pair_$tag/bench.json').read().strip().splitlines()[-1])['ms_per_step'])"
