#!/bin/bash
# GPU box helper: instruction-cache counters per kernel of one bench workload (one rocprofv3 --pmc pass, kernel trace only).
#   tools/pmc_icache.sh <out-name> [bench args...]     e.g. tools/pmc_icache.sh ic_conv --workload heisenberg10x10_conv3x48k3_b4096
# A kernel whose per-iteration code path exceeds the instruction cache shows misses per launch that scale with its
# iteration count (round 5: the unrolled epilogues of the tile GEMMs).
set -uo pipefail
NAME=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$NAME; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 $ROOT/bench.py --steps 3 --warmup 1 --reps 1 --no-cpu-baseline --no-extra --no-timing "$@" > $OUT/a.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in glob.glob(out + '/a/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].rsplit('(', 1)[0].replace('(anonymous namespace)::', '')[:70]
    tot[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
rows = sorted(tot.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0))
with open(out + '/summary.txt', 'w') as o:
  for k, c in rows[:8]:
    n = max(1, len(cnt[k]))
    req, miss, dup = c.get('SQC_ICACHE_REQ', 0) / n, c.get('SQC_ICACHE_MISSES', 0) / n, c.get('SQC_ICACHE_MISSES_DUPLICATE', 0) / n
    o.write('{:70s} launches {:5d}  icache req {:12.0f}  misses {:11.0f} (+dup {:11.0f}) = {:6.2f} %  ifetch {:12.0f}  gui_active/8 {:10.0f}\n'.format(
        k, n, req, miss, dup, 100.0 * (miss + dup) / max(req, 1.0), c.get('SQ_IFETCH', 0) / n, c.get('GRBM_GUI_ACTIVE', 0) / n / 8))
print(open(out + '/summary.txt').read())
PY
rm -rf $OUT/a
