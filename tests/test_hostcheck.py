"""CPU: the host-side planners of the HIP library (cgs_vmc_amd/csrc/plan.hpp -- parameter layouts,
vmc_create's shape / LDS validation, convolution group / band / slice pickers, the sampler's LDS plan,
split-K and XCD block-order maps, SR tile schedules, buffer sizes) under AddressSanitizer +
UndefinedBehaviourSanitizer (SURVEY.md 5 "sanitizers"; GPU sanitizers are not available on this pool).
cgs_vmc_amd/csrc/hostcheck.cpp walks a grid of shapes -- all BASELINE configurations, the limits 512 /
4096 units, 32 filters, kernel 7, 32 x 32 lattices -- and re-derives every index the kernels take from
these plans into real arrays of the planned size."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('g++') is None or shutil.which('make') is None, reason='needs g++ and make')
def test_host_planners_under_asan_and_ubsan():
  p = subprocess.run(['make', '-C', os.path.join(ROOT, 'cgs_vmc_amd', 'csrc'), 'hostcheck'],
                     stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
  out = p.stdout.decode()
  assert p.returncode == 0, out[-4000:]
  assert '-fsanitize=address,undefined' in out
  assert 'ERROR: AddressSanitizer' not in out and 'runtime error' not in out
  m = re.search(r'hostcheck ok: (\d+) shapes \((\d+) rejected by plan_desc, (\d+) convolutional ones on the general path\), '
                r'(\d+) assertions', out)
  assert m, out[-2000:]
  shapes, rejected, general, checks = map(int, m.groups())
  assert shapes > 50000 and 0 < rejected < shapes and 0 < general < shapes and checks > 10 ** 8
