"""Summarises rocprofv3 --pmc counter_collection CSVs per kernel: mean counter value per
dispatch and mean dispatch duration.  Usage: python tools/pmc_summary.py <dir> [<dir> ...]"""
import csv
import glob
import sys
from collections import defaultdict


def summarise(path):
  per = defaultdict(lambda: defaultdict(list))
  dur = defaultdict(dict)
  for f in glob.glob(path + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
      k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
      per[k][r['Counter_Name']].append(float(r['Counter_Value']))
      dur[k][r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
  return per, dur


def main():
  for d in sys.argv[1:]:
    per, dur = summarise(d)
    print('==', d)
    for k in sorted(per, key=lambda k: -sum(dur[k].values())):
      ds = sorted(dur[k].values())
      med = ds[len(ds) // 2]
      print('{:40s} n={:3d} median_us={:9.1f}'.format(k[:40], len(ds), med / 1e3))
      for c, v in sorted(per[k].items()):
        v = sorted(v)
        print('    {:32s} median={:16.0f} mean={:16.0f}'.format(c, v[len(v) // 2], sum(v) / len(v)))


if __name__ == '__main__':
  main()
