"""CPU: static check of the generated code of k_tail16r and k_gemm_ring (tools/check_async_regs.py): between the asm
statement that issues the LDS reads of the next item's operands and the statement that waits for them, no
instruction may name one of the destination registers (the compiler believes them valid from the issue on).  hipcc
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


def test_no_vector_instruction_reads_an_mfma_result_too_early():
  """tools/check_mfma_read_hazard.py on every object of the library (built by __graft_entry__.build() / make): the
  relu is an inline-asm v_max_f32 the compiler's hazard recogniser cannot see; behind an MFMA chain it needs
  vmc_mfma_settle* (the split sampler lost the last k-step of a layer without it; since round 6 the fp32 samplers
  settle too and every MFMA form is strict)."""
  import glob
  objs = sorted(glob.glob(os.path.join(ROOT, 'cgs_vmc_amd', 'csrc', '*.o')))
  if len(objs) < 10:
    import pytest
    pytest.skip('objects not built')
  p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_mfma_read_hazard.py')] + objs,
                     stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
  out = p.stdout.decode()
  assert p.returncode == 0, out[-3000:]
  assert 'violations: 0;' in out
