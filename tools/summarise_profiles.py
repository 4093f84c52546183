"""Summarises gpurun_out/<tag>_prof (tools/collect_profiles.sh <tag> [workload]) into profiles/:
  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (copied)
  <tag>_pmc_summary.txt    per-kernel medians of the PMC passes
  <tag>_traffic.json       HBM bytes per launch of the hot kernels (FETCH_SIZE x 2 per the gfx950
                        correction of MI355X_MICROARCH.md + WRITE_SIZE, KB units -> bytes),
                        MFMA utilisation and clock; bench.py reads it for roofline.traffic
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else 'r4'
SRC = os.path.join(ROOT, 'gpurun_out', TAG + '_prof')
DST = os.path.join(ROOT, 'gpurun_out', TAG + '_summary') if os.environ.get('GRAFT_REPO_ROOT') else os.path.join(ROOT, 'profiles')


def medians(path):
  per = defaultdict(lambda: defaultdict(list))
  dur = defaultdict(dict)
  for f in glob.glob(path + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
      k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
      per[k][r['Counter_Name']].append(float(r['Counter_Value']))
      dur[k][r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
      # the GEMM kernels of the general paths run with several launch shapes in one step (the sampler's B candidates,
      # the local energies' row blocks): also one record per grid, "name @ grid <threads>" (round 6)
      if k.startswith('k_gemm') and r.get('Grid_Size'):
        kg = '{} @ grid {}'.format(k, r['Grid_Size'])
        per[kg][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[kg][r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
  for kg in [k for k in per if ' @ grid ' in k]:      # keep the shapes with at least three launches
    if len(dur[kg]) < 3:
      del per[kg]; del dur[kg]
  out = {}
  for k in per:
    ds = sorted(dur[k].values())
    out[k] = {'n': len(ds), 'median_ns': ds[len(ds) // 2]}
    for c, v in per[k].items():
      v = sorted(v)
      out[k][c] = v[len(v) // 2]
  return out


def main():
  os.makedirs(DST, exist_ok=True)
  for f in glob.glob(SRC + '/stats/**/*kernel_stats.csv', recursive=True):
    shutil.copy(f, os.path.join(DST, TAG + '_kernel_stats.csv'))
  sq = medians(SRC + '/pmc_sq')
  fe = medians(SRC + '/pmc_fetch')
  wr = medians(SRC + '/pmc_write')
  lines = []
  traffic = {}
  for k in sorted(sq, key=lambda k: -sq[k]['median_ns'] * sq[k]['n']):
    s = sq[k]
    lines.append('{}  n={}  median {:.1f} us'.format(k, s['n'], s['median_ns'] / 1e3))
    for c in sorted(s):
      if c not in ('n', 'median_ns'):
        lines.append('    {:32s} {:16.0f}'.format(c, s[c]))
    if fe.get(k, {}).get('SQ_INSTS_VALU_MFMA_MOPS_BF16'):
      lines.append('    {:32s} {:16.0f}  (collected in the FETCH_SIZE pass)'.format('SQ_INSTS_VALU_MFMA_MOPS_BF16', fe[k]['SQ_INSTS_VALU_MFMA_MOPS_BF16']))
    f_kb = fe.get(k, {}).get('FETCH_SIZE')
    w_kb = wr.get(k, {}).get('WRITE_SIZE')
    if f_kb is not None:
      lines.append('    {:32s} {:16.0f}  (KB; x2 on gfx950 for wide reads)'.format('FETCH_SIZE', f_kb))
    if w_kb is not None:
      lines.append('    {:32s} {:16.0f}  (KB)'.format('WRITE_SIZE', w_kb))
    if 'GRBM_GUI_ACTIVE' in s and s['GRBM_GUI_ACTIVE'] > 0:
      cyc = s['GRBM_GUI_ACTIVE'] / 8.0                     # summed over the 8 XCDs
      clock = cyc / s['median_ns']                         # GHz
      util = s.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024.0 * cyc)
      lines.append('    clock {:.3f} GHz   MFMA busy / (1024 SIMD x cycles) = {:.3f}'.format(clock, util))
      if s['median_ns'] < 50e3:
        continue          # GRBM-derived clock is meaningless on very short dispatches
      traffic[k] = {
          'hbm_read_bytes': None if f_kb is None else 2 * 1024 * f_kb,
          'hbm_write_bytes': None if w_kb is None else 1024 * w_kb,
          'median_us': s['median_ns'] / 1e3, 'clock_ghz': clock, 'mfma_util': util, 'launches': s['n'],
      }
  open(os.path.join(DST, TAG + '_pmc_summary.txt'), 'w').write('\n'.join(lines) + '\n')
  # the kernel sources these counters were collected from: bench.py marks the numbers stale when
  # the library it runs was built from other sources
  try:
    sys.path.insert(0, ROOT)
    from cgs_vmc_amd import _hip
    traffic['_collected_at_source_hash'] = _hip.source_hash()
  except Exception:  # pylint: disable=broad-except
    pass
  json.dump(traffic, open(os.path.join(DST, TAG + '_traffic.json'), 'w'), indent=1, sort_keys=True)
  if os.path.exists(SRC + '/bench.json'):
    shutil.copy(SRC + '/bench.json', os.path.join(DST, TAG + '_bench.json'))
  print('\n'.join(lines[:60]))


if __name__ == '__main__':
  main()
