"""CPU: the C-ABI library loads and exports every symbol include/cgsvmc.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
  text = open(os.path.join(ROOT, 'include', 'cgsvmc.h')).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(vmc_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_match_binding_table():
  from cgs_vmc_amd import _hip
  assert _declared() == sorted(_hip.SIGNATURES)


def test_library_loads_and_exports_every_symbol():
  from cgs_vmc_amd import _hip
  if not os.path.exists(_hip.library_path()):
    import __graft_entry__ as g
    g.build()
  lib = _hip.load()
  for name in _declared():
    assert hasattr(lib, name), name
  assert lib.vmc_num_params(100, 256, 3) == 157697


def test_no_gpu_fails_loudly():
  """Without a GPU the product path raises instead of falling back to a CPU path."""
  import torch
  if torch.cuda.is_available():
    pytest.skip('GPU present')
  from cgs_vmc_amd import _hip
  from cgs_vmc_amd.engine import VmcEngine
  with pytest.raises(_hip.HipLibraryError, match='no CPU fallback'):
    VmcEngine(16, 8, 2, 32)
