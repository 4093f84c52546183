#!/bin/bash
# GPU box helper: rocprofv3 --kernel-trace --stats of a short bench run, per-kernel averages on stdout.
#   tools/kstats.sh <out-name> [bench args...]
set -uo pipefail
NAME=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$NAME; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT --output-format csv -- python3 $ROOT/bench.py --steps 10 --warmup 2 --reps 1 --no-cpu-baseline --no-extra --no-timing "$@" > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
  print('{:64s} {:6d} {:10.1f} us {:6.2f} %'.format(r['Name'][:64], int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
