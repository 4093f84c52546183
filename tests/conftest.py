import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session', autouse=True)
def _torch_gpu_context_first(request):
  """GPU runs only: torch creates its HIP context before the first VmcEngine does.  The tests that
  view the library's buffers as torch tensors (parallel.accumulator_tensor) otherwise depend on
  the file order: torch's lazy init after ~140 engine life cycles in the same process was seen to
  report "No HIP GPUs are available".  The product entry points initialise torch first as well
  (bench.py: torch.cuda.set_device; run_training: parallel.init_from_env)."""
  if 'not gpu' in (request.config.getoption('-m') or ''):
    return
  try:
    import torch
    if torch.cuda.device_count() > 0 and torch.cuda.is_available():
      torch.cuda.init()
  except Exception:  # pylint: disable=broad-except
    pass
