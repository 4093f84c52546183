"""CPU: static check of the generated code of k_tail16r (tools/check_async_regs.py): between the asm statement that
issues the LDS reads of the next item's weight fragments and the statement that waits for them, no instruction
may name one of the twelve destination registers (the compiler believes them valid from the issue on).  hipcc
cross-compiles gfx950 here; no GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_instruction_touches_the_in_flight_fragment_registers():
  p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_async_regs.py')], stdout=subprocess.PIPE,
                     stderr=subprocess.STDOUT, timeout=600)
  out = p.stdout.decode()
  assert p.returncode == 0, out[-3000:]
  assert '0 violations' in out
