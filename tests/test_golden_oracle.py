"""CPU: the oracle reproduces the committed golden vectors (tests/golden/vmc_small.npz).
Guards against silent drift of the oracle; the vectors' provenance is stated in
tests/golden/make_golden.py (they do NOT come from the reference's code)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import make_golden  # noqa: E402


def test_oracle_reproduces_golden_file():
  gold = np.load(os.path.join(HERE, 'golden', 'vmc_small.npz'))
  for name in make_golden.CASES:
    fresh = make_golden.build_case(name)
    for k, v in fresh.items():
      g = gold['{}/{}'.format(name, k)]
      if np.issubdtype(np.asarray(v).dtype, np.floating):
        np.testing.assert_allclose(g, v, rtol=1e-12, atol=1e-12, err_msg='{}/{}'.format(name, k))
      else:
        np.testing.assert_array_equal(g, v, err_msg='{}/{}'.format(name, k))


def test_oracle_reproduces_rbm_golden_file():
  gold = np.load(os.path.join(HERE, 'golden', 'rbm_small.npz'))
  for name in make_golden.RBM_CASES:
    fresh = make_golden.build_rbm_case(name)
    for k, v in fresh.items():
      g = gold['{}/{}'.format(name, k)]
      if np.issubdtype(np.asarray(v).dtype, np.floating):
        np.testing.assert_allclose(g, v, rtol=1e-12, atol=1e-12, err_msg='{}/{}'.format(name, k))
      else:
        np.testing.assert_array_equal(g, v, err_msg='{}/{}'.format(name, k))


def test_oracle_reproduces_conv_golden_file():
  gold = np.load(os.path.join(HERE, 'golden', 'conv_small.npz'))
  for name in make_golden.CONV_CASES:
    fresh = make_golden.build_conv_case(name)
    for k, v in fresh.items():
      g = gold['{}/{}'.format(name, k)]
      if np.issubdtype(np.asarray(v).dtype, np.floating):
        np.testing.assert_allclose(g, v, rtol=1e-11, atol=1e-11, err_msg='{}/{}'.format(name, k))
      else:
        np.testing.assert_array_equal(g, v, err_msg='{}/{}'.format(name, k))


def test_oracle_reproduces_wide_golden_files():
  """Round-3 fixtures: more than 16 filters / cos (conv_wide.npz), rbm beyond 256 units (rbm_wide.npz)."""
  for fname, cases, build, tol in (
      ('conv_wide.npz', make_golden.CONV_WIDE_CASES, lambda n: make_golden.build_conv_case(n, make_golden.CONV_WIDE_CASES, 51), 1e-11),
      ('rbm_wide.npz', make_golden.RBM_WIDE_CASES, lambda n: make_golden.build_rbm_case(n, make_golden.RBM_WIDE_CASES), 1e-12)):
    gold = np.load(os.path.join(HERE, 'golden', fname))
    for name in cases:
      for k, v in build(name).items():
        g = gold['{}/{}'.format(name, k)]
        if np.issubdtype(np.asarray(v).dtype, np.floating):
          np.testing.assert_allclose(g, v, rtol=tol, atol=tol, err_msg='{}/{}'.format(name, k))
        else:
          np.testing.assert_array_equal(g, v, err_msg='{}/{}'.format(name, k))
