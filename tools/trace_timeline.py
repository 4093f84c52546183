"""Prints the kernel timeline (start / end relative to the first kernel, us) of a few steps from a
rocprofv3 --kernel-trace csv: which kernels overlap, where the gaps are.
Usage: python tools/trace_timeline.py <kernel_trace.csv> [first_row [n_rows]]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
t0 = int(rows[first]['Start_Timestamp'])
prev_end = t0
for r in rows[first:first + count]:
  s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
  name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:44]
  print('%-46s q%-3s start %9.1f  dur %8.1f  gap_after_prev_end %7.1f' % (
      name, r.get('Queue_Id', '?'), (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
  prev_end = max(prev_end, e)
