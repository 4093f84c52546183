"""GPU: the overtaking sampler (CGS_VMC_OVERLAP=2: the launch that follows an accumulate waits only for the event
recorded when that accumulate STARTED) against stream order (CGS_VMC_OVERLAP=0) through the public op-by-op API.
ADVICE r4: the LogOverlapITSWO accumulate refreshes the supervisor's cache with a zero-step pass of the sampler
kernel, which writes its chain copy into the buffer the next sampler launch writes too; a following vmc_mc_steps
must therefore wait for it (no token) -- chains, caches and accumulators have to be the same bits either way."""
import numpy as np
import pytest

from oracle import vmc_oracle as vo

pytestmark = pytest.mark.gpu


def _run(monkeypatch, overlap, mode):
  from cgs_vmc_amd import _hip
  from cgs_vmc_amd.engine import VmcEngine
  monkeypatch.setenv('CGS_VMC_OVERLAP', overlap)
  n, h, L, b = 100, 256, 3, 4096          # one 16-chain tile per CU: two sampler launches exceed the chip
  rng = np.random.default_rng(11)
  theta = vo.init_params(n, h, L, rng) + (0.03 * rng.standard_normal(vo.num_params(n, h, L))).astype(np.float32)
  cfg = vo.random_configurations(n, b, np.random.RandomState(12))
  eng = VmcEngine(n, b, L, h, seed=2024)
  eng.set_params(theta); eng.set_configs(cfg); eng.set_bonds(vo.torus_bonds(10, 10, False), -1.0, 1.0)
  eng.transfer_params()
  eng.mc_steps(3, want_accepted=False)     # leaves a valid psi cache: the next accumulate hands out its token
  out = []
  eng.reset_accumulators()
  for _ in range(6):
    eng.accumulate(mode, 0.12)
    eng.mc_steps(7, want_accepted=False)   # directly behind the accumulate, no parameter update in between
    out.append((eng.get_configs(), eng.amplitude()[0], eng.local_energy()[0]))
  acc = eng.get_accumulators()
  eng.close()
  return out, acc


@pytest.mark.parametrize('mode', [1, 0], ids=['log_overlap_itswo', 'energy_gradient'])
def test_overtaking_sampler_equals_stream_order(monkeypatch, mode):
  ref, acc_ref = _run(monkeypatch, '0', mode)
  got, acc = _run(monkeypatch, '2', mode)
  for k, (r, g) in enumerate(zip(ref, got)):
    np.testing.assert_array_equal(r[0], g[0], err_msg='chains after round {}'.format(k))
    np.testing.assert_array_equal(r[1], g[1], err_msg='cached logits after round {}'.format(k))
    np.testing.assert_array_equal(r[2], g[2], err_msg='local energies after round {}'.format(k))
  np.testing.assert_array_equal(acc_ref, acc)
