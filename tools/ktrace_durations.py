"""Per-dispatch durations of the kernels whose name contains <filter>, from rocprofv3 kernel_trace CSVs.
Usage: python tools/ktrace_durations.py <dir> <filter>"""
import csv
import glob
import sys
from collections import defaultdict

d, flt = sys.argv[1], sys.argv[2]
out = defaultdict(list)
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    if flt in r['Kernel_Name']:
      out[r['Kernel_Name'].replace('(anonymous namespace)::', '')[:60]].append(
          (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in out.items():
  print(k, len(v), 'us:', ' '.join('%.0f' % x for x in v[-12:]))
