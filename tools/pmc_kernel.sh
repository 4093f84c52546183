#!/bin/bash
# GPU box helper: two rocprofv3 --pmc passes (kernel trace only) of one bench workload, summed per kernel name.
#   tools/pmc_kernel.sh <out-name> <kernel-substring> [bench args...]
# e.g. tools/pmc_kernel.sh r5_split_ring k_tail16 --workload heisenberg10x10_fc3x256_b4096_split3xbf16
set -uo pipefail
NAME=$1; KSUB=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$NAME; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --steps 10 --warmup 2 --reps 1 --no-cpu-baseline --no-extra --no-timing $*"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- $B > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU --output-format csv -d $OUT/b -- $B > $OUT/b.log 2>&1
python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, sys, collections
out, ksub = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in glob.glob(out + '/*/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if ksub not in k: continue
    k = k.split('(')[0][:60]
    tot[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k].add((f, r['Dispatch_Id']))
with open(out + '/summary.txt', 'w') as o:
  for k in tot:
    n = max(1, len(cnt[k]) // 2)
    o.write('{}  (dispatches per pass ~{})\n'.format(k, n))
    for c in sorted(tot[k]):
      o.write('    {:32s} {:16.0f}\n'.format(c, tot[k][c] / n))
print(open(out + '/summary.txt').read())
PY
rm -rf $OUT/a $OUT/b
