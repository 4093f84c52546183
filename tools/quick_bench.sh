#!/bin/bash
# GPU box helper (through gpurun): a subset of the GPU tests, one bench line, and a rocprofv3 kernel-trace
# summary of a short bench run.  Usage: tools/quick_bench.sh <tag> [pytest files ...]
set -uo pipefail
TAG=$1; shift
OUT=gpurun_out/r4/$TAG
mkdir -p $OUT
if [ $# -gt 0 ]; then
  timeout -k 10 900 python -m pytest "$@" -q -m gpu -x > $OUT/tests.log 2>&1; echo rc=$? >> $OUT/tests.log
  tail -4 $OUT/tests.log
fi
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["repetitions_ms_per_step"])
print({k: round(v["ms_avg"] * 1e3, 1) for k, v in d["kernels"].items()})
PY
ROOT=$(pwd)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o p -- python3 $ROOT/bench.py --steps 20 --warmup 5 --reps 2 --no-cpu-baseline --no-extra --no-timing > /dev/null 2>&1)
F=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
[ -n "$F" ] && cut -d, -f1-4 "$F" | sed 's/(.*)//' | cut -c1-110 | head -14
