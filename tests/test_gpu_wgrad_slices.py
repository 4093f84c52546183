"""GPU: k_wgrad's split-K hand-over (sc1 partial tiles -> arrival ticket -> the last arriver folds slices 0..S-1 in
that fixed order) for EVERY slice count 1..32 (CGS_VMC_WGRAD_SLICES, read per launch): over repeated launches the
accumulators of one slice count are the same bits whoever arrived last, and every slice count agrees with the
unsplit sum (S = 1: no workspace, no ticket) within the fp32 re-association of a 4096-term sum."""
import numpy as np
import pytest

from oracle import vmc_oracle as vo

pytestmark = pytest.mark.gpu


def test_every_slice_count_is_deterministic_and_agrees_with_the_unsplit_sum(monkeypatch):
  from cgs_vmc_amd import _hip
  from cgs_vmc_amd.engine import VmcEngine
  n, h, L, b = 100, 256, 3, 4096          # config 3: 40 tiles x S slices over all eight XCDs
  rng = np.random.default_rng(21)
  theta = vo.init_params(n, h, L, rng) + (0.03 * rng.standard_normal(vo.num_params(n, h, L))).astype(np.float32)
  eng = VmcEngine(n, b, L, h, seed=2024)
  eng.set_params(theta); eng.set_configs(vo.random_configurations(n, b, np.random.RandomState(22)))
  eng.set_bonds(vo.torus_bonds(10, 10, False), -1.0, 1.0)
  eng.mc_steps(5, want_accepted=False)

  def acc_once():
    eng.reset_accumulators()
    eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
    return eng.get_accumulators()

  monkeypatch.setenv('CGS_VMC_WGRAD_SLICES', '1')
  ref = acc_once()
  np.testing.assert_array_equal(ref, acc_once())
  scale = np.abs(ref).max()
  worst = 0.0
  for s in range(2, 33):
    monkeypatch.setenv('CGS_VMC_WGRAD_SLICES', str(s))
    first = acc_once()
    for _ in range(5):                      # another arrival order every launch, the same bits
      np.testing.assert_array_equal(first, acc_once(), err_msg='slices {}'.format(s))
    worst = max(worst, float(np.abs(first - ref).max() / scale))
  assert worst <= 2e-6, worst               # fp32 sums of 4096 terms in another order
  eng.close()
