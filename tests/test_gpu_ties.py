"""The first-index tie rule of tf.argmax / tf.argmin in the exchange proposal
(graph_builders.py:62-65) on the PRODUCTION sampler kernels.

tests/golden/tie_events.json (tests/golden/gen_tie_events.py) lists (chain id, step) pairs of the
seed-2024 Philox stream in which the largest site uniform of a chain occurs at two sites.  With both
sites up (down), argmax (argmin) of s*u must pick the smaller index.  All parameters are zero, so
psi is constant and every proposal is accepted (1 > sqrt(u)): the chains after one mc_step show the
proposal of every chain, which must equal the oracle's bit for bit.  Covers the integer-key
proposals of the hand-over variants (256 units with W1 in LDS, 129..256 sites with their own
hand-over area, 512 units, RBM) and the float formulation of the 4-wave variants (128 units).
"""
import json
import os

import numpy as np
import pytest

from oracle import vmc_oracle as vo

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, 'golden', 'tie_events.json')) as f:
  _TIES = json.load(f)
SEED = _TIES['seed']
EVENTS = _TIES['events']

# (ansatz, layer size, num layers) per lattice size
VARIANTS = {
    100: [('fully_connected', 256, 3), ('fully_connected', 256, 2), ('fully_connected', 512, 3),
          ('fully_connected', 128, 3), ('rbm', 256, 2)],
    256: [('fully_connected', 256, 3), ('fully_connected', 256, 6)],
}
CASES = [(e, v, s) for e in EVENTS for v in VARIANTS[e['n_sites']] for s in (1.0, -1.0)]


def _row_with(n, sites, spin, rng):
  """Sz = 0 row whose `sites` all carry `spin`."""
  rest = [i for i in range(n) if i not in sites]
  rng.shuffle(rest)
  row = np.empty(n, np.float32)
  row[list(sites)] = spin
  n_same = n // 2 - len(sites)
  row[rest[:n_same]] = spin
  row[rest[n_same:]] = -spin
  return row


@pytest.mark.parametrize('event,variant,spin', CASES,
                         ids=['n{}c{}s{}-{}{}x{}-{}'.format(e['n_sites'], e['chain'], e['step'], v[0][:2], v[2], v[1],
                                                          'up' if s > 0 else 'dn') for e, v, s in CASES])
def test_tied_uniforms_pick_the_first_site(event, variant, spin):
  from cgs_vmc_amd.engine import VmcEngine
  n, chain, step, sites = event['n_sites'], event['chain'], event['step'], event['sites']
  ansatz, h, L = variant
  b, k = 32, 5                       # the chain sits at row 5 of the batch
  offset = chain - k
  rng = np.random.default_rng(chain)
  cfg = vo.random_configurations(n, b, np.random.RandomState(step))
  cfg[k] = _row_with(n, sites, np.float32(spin), rng)
  assert cfg[k].sum() == 0

  ids = np.arange(b, dtype=np.uint32) + np.uint32(offset)
  u_sites, u_acc = vo.step_uniforms(SEED, ids, step, n)
  assert u_sites[k, sites[0]] == u_sites[k, sites[1]] == u_sites[k].max()   # the fixture is what it says
  i_up, i_dn = vo.propose_exchange(cfg, u_sites)
  assert (i_up[k] if spin > 0 else i_dn[k]) == min(sites)                    # oracle: first index
  want = cfg.copy()
  rows = np.arange(b)
  want[rows, i_dn] = 1.0
  want[rows, i_up] = -1.0
  assert (u_acc < 1.0).all()

  n_par = vo.ANSATZ[ansatz][4](n, h, L)
  eng = VmcEngine(n, b, L, h, chain_offset=offset, seed=SEED, ansatz=ansatz)
  eng.set_params(np.zeros(n_par, np.float32))      # constant psi: every proposal is accepted
  eng.set_configs(cfg)
  eng.step_counter = step
  accepted = eng.mc_steps(1)
  got = eng.get_configs()
  eng.close()
  assert accepted == b
  np.testing.assert_array_equal(got, want)
