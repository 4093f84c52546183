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


def test_no_vector_instruction_reads_a_bf16_mfma_result_too_early():
  """tools/check_mfma_read_hazard.py on the objects of the two 3 x bf16 split kernels (built by
  __graft_entry__.build() / make): the relu is an inline-asm v_max_f32 the compiler's hazard recogniser cannot see;
  behind a v_mfma_f32_16x16x32_bf16 chain it needs vmc_mfma_settle (the split sampler lost the last k-step of a
  layer without it)."""
  objs = [os.path.join(ROOT, 'cgs_vmc_amd', 'csrc', n) for n in ('sweep_split.o', 'tail_split.o')]
  if not all(os.path.exists(o) for o in objs):
    import pytest
    pytest.skip('objects not built')
  p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_mfma_read_hazard.py')] + objs,
                     stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
  out = p.stdout.decode()
  assert p.returncode == 0, out[-3000:]
  assert 'violations: 0;' in out
