#!/bin/bash
# GPU box helper (round 6): what k_sr_rowdot / k_sr_wsum wait on -- a PMC pass of the SR CG loop at
# config 3 (tools/sr_bench.py 50 20: 204,800 stored samples, 20 CG iterations), counters in their own run (no stats).
#   usage: tools/collect_sr_pmc.sh <tag>      -> gpurun_out/<tag>_sr_pmc_summary.txt
set -uo pipefail
T=${1:-r6}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${T}_sr_pmc; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $ROOT/tools/sr_bench.py 50 8 > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/fetch -- python3 $ROOT/tools/sr_bench.py 50 8 > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/tools/sr_bench.py 50 8 > $OUT/write.log 2>&1
python3 - "$OUT" "$ROOT/gpurun_out/${T}_sr_pmc_summary.txt" <<'PY'
import csv, glob, sys, collections
src, dst = sys.argv[1], sys.argv[2]
per = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(dict)
for f in glob.glob(src + '/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
    per[k][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[k][(f, r['Dispatch_Id'])] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
lines = []
for k in sorted(per, key=lambda k: -sum(dur[k].values())):
  ds = sorted(dur[k].values())
  if ds[len(ds) // 2] < 20e3: continue
  lines.append('{}  n={}  median {:.1f} us'.format(k, len(ds), ds[len(ds) // 2] / 1e3))
  c = {n: sorted(v)[len(v) // 2] for n, v in per[k].items()}
  for n in sorted(c): lines.append('    {:32s} {:16.0f}'.format(n, c[n]))
  if c.get('GRBM_GUI_ACTIVE'):
    cyc = c['GRBM_GUI_ACTIVE'] / 8.0
    lines.append('    MFMA busy / (1024 SIMD x cycles) = {:.3f};  wait_any / wave_cycles = {:.3f};  HBM read {:.1f} MB (2 x FETCH_SIZE KB) write {:.1f} MB'.format(
        c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024.0 * cyc), c.get('SQ_WAIT_ANY', 0) / max(1.0, c.get('SQ_WAVE_CYCLES', 1)),
        2 * c.get('FETCH_SIZE', 0) / 1024.0, c.get('WRITE_SIZE', 0) / 1024.0))
open(dst, 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines[:70]))
PY
rm -rf $OUT/sq $OUT/fetch $OUT/write
