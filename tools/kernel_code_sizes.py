#!/usr/bin/env python3
"""Instruction bytes of every kernel in the built objects (cgs_vmc_amd/csrc/*.o), largest first.  A kernel whose
straight-line code is far beyond the 64 KB instruction cache walks through it one miss after the other (round 5:
the unrolled element epilogue of the tile GEMMs).  usage: python tools/kernel_code_sizes.py [n]"""
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from check_mfma_read_hazard import LLVM, gfx950_objects  # noqa: E402


def main():
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
  rows = []
  with tempfile.TemporaryDirectory() as tmp:
    for o in sorted(glob.glob(os.path.join(ROOT, 'cgs_vmc_amd', 'csrc', '*.o'))):
      d = os.path.join(tmp, os.path.basename(o))
      os.mkdir(d)
      for elf in gfx950_objects(o, d):
        s = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '-s', '--wide', elf], stdout=subprocess.PIPE,
                           stderr=subprocess.DEVNULL).stdout.decode()
        for line in s.splitlines():
          p = line.split()
          if len(p) >= 8 and p[3] == 'FUNC':
            rows.append((int(p[2]), os.path.basename(o), p[7]))
  rows.sort(reverse=True)
  for size, obj, name in rows[:n]:
    dem = subprocess.run(['c++filt', name], stdout=subprocess.PIPE).stdout.decode().strip()
    print('{:8d}  {:22s} {}'.format(size, obj, dem[:120]))


if __name__ == '__main__':
  main()
