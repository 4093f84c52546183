#!/usr/bin/env python3
"""Static check of the asynchronous register hand-overs of k_tail16r (cgs_vmc_amd/csrc/tail_split.hip, LDS_STEP)
and k_gemm_ring (cgs_vmc_amd/csrc/grad.hip, GR_STEP: two ds_read_b128 + two ds_read2st64_b32 per statement).

An LDS_STEP statement issues three ds_read_b128 into the register set of the NEXT item and only the following
LDS_STEP waits for them (s_waitcnt lgkmcnt(0) at its head).  The compiler believes the destination registers valid
from the issuing statement on, so nothing in the generated code may touch them in between: a copy would move stale
data, and anything the register allocator parked there would be overwritten when the reads land.  This script
compiles tail_split.hip to assembly (hipcc -S, gfx950; no GPU needed) and walks every such window of both
instantiations: any instruction between the issue and the next wait that names one of the twelve in-flight
registers -- as a source or as a destination -- fails the check.  tests/test_async_regs.py runs it in the CPU suite.

  python tools/check_async_regs.py [--keep file.s]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'cgs_vmc_amd', 'csrc')
# source, kernel-name fragment of the mangled symbol, instantiations expected, in-flight windows expected per instantiation
TARGETS = [('tail_split.hip', 'k_tail16r', 2, 2 * 128), ('grad.hip', 'k_gemm_ring', 4, 17)]
READS = ('ds_read_b128', 'ds_read2st64_b32')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')

REG_RANGE = re.compile(r'\b([va])\[(\d+):(\d+)\]')
REG_ONE = re.compile(r'\b([va])(\d+)\b')


def regs_of(text):
  out = set()
  for kind, lo, hi in REG_RANGE.findall(text):
    out.update((kind, i) for i in range(int(lo), int(hi) + 1))
  text = REG_RANGE.sub(' ', text)
  for kind, i in REG_ONE.findall(text):
    out.add((kind, int(i)))
  return out


def assembly(src):
  with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, 'out.s')
    cmd = [HIPCC, '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '--cuda-device-only', '-S',
           os.path.join(CSRC, src), '-o', out]
    subprocess.run(cmd, check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
    return open(out).read()


def kernels(text, fragment):
  cur, name = [], None
  for line in text.splitlines():
    m = re.match(r'^(_Z\w*' + fragment + r'\w*):', line)
    if m:
      name, cur = m.group(1), []
      continue
    if name:
      cur.append(line)
      if 's_endpgm' in line:
        yield name, cur
        name = None


def check(name, lines):
  """Returns (windows, violations)."""
  windows, bad = 0, []
  inflight, issued_at = None, None
  i = 0
  n = len(lines)
  while i < n:
    line = lines[i].strip()
    if line.startswith(';;#ASMSTART'):
      j = i + 1
      block = []
      while j < n and not lines[j].strip().startswith(';;#ASMEND'):
        block.append(lines[j].strip())
        j += 1
      reads = [b for b in block if b.startswith(READS)]
      waits = any(b.startswith('s_waitcnt') and 'lgkmcnt(0)' in b for b in block)
      if waits and inflight is not None:
        inflight = None                                  # the window closes at this block's wait
      if reads:
        # a block that reads ring fragments: its destinations are in flight until the next waiting block;
        # a wait in the SAME block stands in front of the reads (LDS_STEP), so it does not close this window
        dst = set()
        for b in reads:
          dst |= regs_of(b.split(',')[0])
        wait_after = False
        for b in block:
          if b.startswith(READS):
            wait_after = False
          elif b.startswith('s_waitcnt') and 'lgkmcnt(0)' in b:
            wait_after = True
        if not wait_after:
          inflight, issued_at = dst, i
          windows += 1
      i = j + 1
      continue
    if inflight is not None and line and not line.startswith(';') and not line.startswith('.'):
      code = line.split(';')[0]
      touched = regs_of(code) & inflight
      if touched:
        bad.append((name, i + 1, issued_at + 1, code.strip(), sorted(touched)))
    i += 1
  return windows, bad


def main():
  keep = sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] == '--keep' else None
  total, bad, seen = 0, [], 0
  for src, fragment, n_inst, n_windows in TARGETS:
    text = assembly(src)
    if keep:
      open(keep + '.' + src + '.s', 'w').write(text)
    here, windows = 0, 0
    for name, lines in kernels(text, fragment):
      here += 1
      w, b = check(name, lines)
      windows += w
      bad += b
    if here != n_inst or windows < n_inst * n_windows:
      print('check_async_regs: {}: expected {} instantiation(s) of {} with >= {} windows each, saw {} kernels, {} windows'
            .format(src, n_inst, fragment, n_windows, here, windows))
      return 2
    seen += here
    total += windows
  for name, line, issued, code, regs in bad[:20]:
    print('{}: line {} touches {} in flight since the asm block at line {}: {}'.format(name, line, regs, issued, code))
  print('check_async_regs: {} kernels, {} in-flight windows, {} violations'.format(seen, total, len(bad)))
  return 1 if bad else 0


if __name__ == '__main__':
  sys.exit(main())
