"""Durations of the kernels whose name contains <filter>, grouped by grid size, from rocprofv3 kernel_trace CSVs.
Usage: python tools/ktrace_by_grid.py <dir> <filter>"""
import csv
import glob
import sys
from collections import defaultdict

d, flt = sys.argv[1], sys.argv[2]
out = defaultdict(list)
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    if flt in r['Kernel_Name']:
      key = (r['Kernel_Name'].replace('(anonymous namespace)::', '')[:40], int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])))
      out[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k in sorted(out):
  v = sorted(out[k])
  print('{:42s} workgroups {:7d}  n {:5d}  median {:9.1f} us  min {:9.1f}  max {:9.1f}'.format(k[0], k[1], len(v), v[len(v) // 2], v[0], v[-1]))
