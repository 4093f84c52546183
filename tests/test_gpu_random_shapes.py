"""GPU parity on seeded random shapes (both ansatz types): lattice size, hidden width, depth,
batch and bond graph with per-bond couplings are drawn at random, so padding (H not a multiple
of 16 / 64), ragged batches, odd N, tiny H and irregular bond lists all get exercised.
Tolerances as in tests/test_gpu_engine.py."""
import numpy as np
import pytest

from oracle import vmc_oracle as vo

pytestmark = pytest.mark.gpu


def _draw(seed, big=False):
  """big: hidden widths beyond the register-resident kernels (the padded 384 / 512-unit fused path
  and the general path of csrc/wide.hip) and lattices of up to 300 sites (beyond the prefetched
  Philox blocks and, at 256 units, beyond the W1-in-LDS sampler)."""
  rng = np.random.default_rng((7000 if big else 1000) + seed)
  ansatz = 'rbm' if seed % 3 == 2 else 'fully_connected'
  if big:
    n = int(rng.choice([int(rng.integers(4, 61)), int(rng.integers(100, 301))]))
    h = int(rng.choice([64, 256, 257, 300, 384, 400, 500, 512, 513, 700]))
    L = int(rng.integers(0, 3)) if ansatz == 'rbm' else int(rng.integers(1, 4))
    b = int(rng.integers(1, 81))
    n_b = int(rng.integers(1, n + 20))
  else:
    n = int(rng.integers(4, 61))
    h = int(rng.choice([1, 3, 16, 17, 40, 64, 65, 100, 128, 129, 200, 255, 256]))
    L = int(rng.integers(0, 4)) if ansatz == 'rbm' else int(rng.integers(1, 5))
    b = int(rng.integers(1, 151))
    n_b = int(rng.integers(1, 3 * n))
  bonds = []
  while len(bonds) < n_b:
    i, j = (int(x) for x in rng.integers(0, n, 2))
    if i != j:
      bonds.append((i, j))
  jx = rng.uniform(-1.5, 1.5, n_b).astype(np.float32)
  jz = rng.uniform(-1.0, 1.5, n_b).astype(np.float32)
  return ansatz, n, h, L, b, bonds, jx, jz, rng


@pytest.mark.parametrize('seed', range(30))
def test_random_wide_or_large_shape_matches_oracle(seed):
  _check_random_shape(seed, True)


@pytest.mark.parametrize('seed', range(60))
def test_random_shape_matches_oracle(seed):
  _check_random_shape(seed, False)


def _check_random_shape(seed, big):
  from cgs_vmc_amd import _hip
  from cgs_vmc_amd.engine import VmcEngine
  ansatz, n, h, L, b, bonds, jx, jz, rng = _draw(seed, big)
  rbm = ansatz == 'rbm'
  theta = (vo.rbm_init_params if rbm else vo.init_params)(n, h, L, rng)
  if big:
    # Sonnet's default initialisation puts the logit of a 500..700-unit network at ~ 100: exp(logit + 10)
    # overflows the reference's own fp32 psi and fp32 logits resolve psi'/psi to 1e-5 only, beyond the
    # stated local-energy tolerance.  Scaled-down weights keep these cases inside the fp32 domain.
    theta = (0.3 * theta + 0.01 * rng.standard_normal(theta.size)).astype(np.float32)
  else:
    theta = (theta + 0.05 * rng.standard_normal(theta.size)).astype(np.float32)
  logit_fn = vo.rbm_logit if rbm else vo.fc_logit
  psi_fn = vo.rbm_psi if rbm else vo.fc_psi
  cfg = vo.random_configurations(n, b, np.random.RandomState(seed))
  eng = VmcEngine(n, b, L, h, seed=11, ansatz=ansatz)
  eng.set_params(theta); eng.set_configs(cfg); eng.set_bonds(bonds, jx, jz)
  tag = (ansatz, n, h, L, b, len(bonds))

  def close(a, ref, rel):
    a = np.asarray(a, np.float64); ref = np.asarray(ref, np.float64)
    err = np.abs(a - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() <= rel, (tag, float(err.max()))

  close(eng.amplitude()[0], logit_fn(theta, cfg, h, L, dtype=np.float64), 2e-5)
  amp = lambda c: psi_fn(theta, c, h, L, dtype=np.float64)
  close(eng.local_energy()[0], vo.local_value(amp, cfg, bonds, jx, jz, dtype=np.float64), 3e-4)
  # gradient accumulators
  acc = vo.Accumulators(theta.size, np.float64)
  vo.energy_gradient_accumulate(acc, theta, cfg, bonds, jx, jz, -10.0, h, L, np.float64,
                                ansatz=ansatz)
  eng.reset_accumulators()
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  got = eng.get_accumulators()
  p = theta.size
  for g, r in ((got[:p], acc.g1_total), (got[p:2 * p], acc.g2_total)):
    assert np.abs(g - r).max() < 2e-3 * np.abs(r).max() + 2e-4, tag
  # one injected Metropolis step + exact cache afterwards
  u_sites, u_acc = vo.step_uniforms(5, np.arange(b), seed, n)
  i_up, i_dn = vo.propose_exchange(cfg, u_sites)
  _, acc_ref, ratios = vo.mc_step(amp, cfg, i_up, i_dn, u_acc)
  mask = eng.mc_step_injected(i_up, i_dn, u_acc)
  band = np.abs(ratios - np.sqrt(u_acc.astype(np.float64))) < 1e-4 * np.maximum(ratios, 1e-30)
  assert np.array_equal(mask[~band], acc_ref[~band]), tag
  got_cfg = eng.get_configs()
  close(eng.amplitude()[0], logit_fn(theta, got_cfg, h, L, dtype=np.float64), 2e-5)
  # a short free-running sweep keeps Sz and the cache exact
  eng.mc_steps(2 * n)
  got_cfg = eng.get_configs()
  assert (np.abs(got_cfg) == 1).all() and (got_cfg.sum(1) == cfg.sum(1)).all(), tag
  close(eng.amplitude()[0], logit_fn(theta, got_cfg, h, L, dtype=np.float64), 2e-5)
  eng.close()


@pytest.mark.parametrize('seed', range(20))
def test_random_shape_itswo_and_sr(seed):
  """LogOverlapITSWO accumulators and the SR matrix-vector product on small random shapes
  (dense S is affordable there)."""
  from cgs_vmc_amd import _hip
  from cgs_vmc_amd.engine import VmcEngine
  rng = np.random.default_rng(5000 + seed)
  rbm = seed % 2 == 1
  ansatz = 'rbm' if rbm else 'fully_connected'
  n = int(rng.integers(4, 21)); h = int(rng.choice([2, 7, 16, 33, 48]))
  L = int(rng.integers(0, 3)) if rbm else int(rng.integers(1, 4))
  b = int(rng.integers(2, 61))
  bonds = [(i, (i + 1) % n) for i in range(n)] + [(0, n // 2)]
  theta = (vo.rbm_init_params if rbm else vo.init_params)(n, h, L, rng)
  theta = (theta + 0.05 * rng.standard_normal(theta.size)).astype(np.float32)
  theta_w = (theta + 0.02 * rng.standard_normal(theta.size)).astype(np.float32)
  cfg = vo.random_configurations(n, b, np.random.RandomState(seed))
  tag = (ansatz, n, h, L, b)
  eng = VmcEngine(n, b, L, h, seed=3, ansatz=ansatz)
  eng.set_bonds(bonds, -1.0, 1.0)
  # ITSWO: omega = theta_w frozen, psi = theta
  eng.set_params(theta_w); eng.set_configs(cfg)
  eng.transfer_params()
  eng.set_params(theta); eng.set_shift(-9.5)
  acc = vo.Accumulators(theta.size, np.float64)
  vo.log_overlap_accumulate(acc, theta, theta_w, cfg, bonds, -1.0, 1.0, -9.5, -10.0, 0.12, h, L,
                            np.float64, ansatz=ansatz)
  eng.reset_accumulators()
  eng.accumulate(_hip.VMC_MODE_LOG_OVERLAP_ITSWO, 0.12)
  g_ref = vo.log_overlap_gradient(acc)
  g = eng.get_gradient(_hip.VMC_MODE_LOG_OVERLAP_ITSWO)
  assert np.abs(g - g_ref).max() < 2e-3 * np.abs(g_ref).max() + 2e-4, tag
  # SR matvec over two recorded batches
  eng.set_shift(-10.0)
  eng.sr_reserve(2)
  eng.reset_accumulators()
  cfgs, elocs = [], []
  for k in range(2):
    c = vo.random_configurations(n, b, np.random.RandomState(100 + seed + k))
    eng.set_configs(c); eng.accumulate(0)
    cfgs.append(c); elocs.append(eng.local_energy()[0])
  grads = vo.rbm_per_sample_logit_grads if rbm else vo.per_sample_logit_grads
  o = grads(theta, np.concatenate(cfgs, 0), h, L)
  s_mat, _ = vo.sr_system(o, np.concatenate(elocs, 0).astype(np.float64))
  v = rng.standard_normal(theta.size).astype(np.float32)
  ref = s_mat @ v.astype(np.float64) + 0.01 * v
  got = eng.sr_debug_matvec(v, 0.01)
  assert np.abs(got - ref).max() <= 3e-4 * np.abs(ref).max(), tag
  eng.close()


@pytest.mark.parametrize('seed', range(24))
def test_random_shape_with_a_random_hidden_activation(seed):
  """fully_connected over the hidden activations of layers.NONLINEARITIES that keep a random network
  inside fp32 (relu, tanh, sigmoid, cos, identity) at widths on both sides of the 256-unit border of the
  register-resident kernels (cos beyond it since round 3): logits, local energies, gradient sums, one
  injected step."""
  from cgs_vmc_amd import _hip
  from cgs_vmc_amd.engine import VmcEngine
  rng = np.random.default_rng(9000 + seed)
  nonlin = ['relu', 'tanh', 'sigmoid', 'cos', 'identity'][seed % 5]
  n = int(rng.integers(6, 41))
  h = int(rng.choice([24, 100, 256, 272, 384, 448, 512]))
  L = int(rng.integers(1, 4))
  b = int(rng.integers(2, 61))
  n_b = int(rng.integers(2, 2 * n))
  bonds = []
  while len(bonds) < n_b:
    i, j = (int(x) for x in rng.integers(0, n, 2))
    if i != j:
      bonds.append((i, j))
  jx = rng.uniform(-1.5, 1.5, n_b).astype(np.float32)
  jz = rng.uniform(-1.0, 1.5, n_b).astype(np.float32)
  theta = vo.init_params(n, h, L, rng)
  theta = ((0.3 if h > 256 else 1.0) * theta + 0.02 * rng.standard_normal(theta.size)).astype(np.float32)
  cfg = vo.random_configurations(n, b, np.random.RandomState(seed))
  eng = VmcEngine(n, b, L, h, seed=11, nonlinearity=nonlin)
  assert eng.kernel_path() == (1 if h > 256 else 0)
  eng.set_params(theta); eng.set_configs(cfg); eng.set_bonds(bonds, jx, jz)
  tag = (nonlin, n, h, L, b, n_b)
  kw = dict(nonlinearity=nonlin, dtype=np.float64)

  def close(a, ref, rel):
    a = np.asarray(a, np.float64); ref = np.asarray(ref, np.float64)
    err = np.abs(a - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() <= rel, (tag, float(err.max()))

  close(eng.amplitude()[0], vo.fc_logit(theta, cfg, h, L, **kw), 2e-5)
  amp = lambda c: vo.fc_psi(theta, c, h, L, **kw)
  close(eng.local_energy()[0], vo.local_value(amp, cfg, bonds, jx, jz, dtype=np.float64), 3e-4)
  acc = vo.Accumulators(theta.size, np.float64)
  vo.energy_gradient_accumulate(acc, theta, cfg, bonds, jx, jz, -10.0, h, L, np.float64, nonlinearity=nonlin)
  eng.reset_accumulators()
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  got = eng.get_accumulators()
  p = theta.size
  for g, r in ((got[:p], acc.g1_total), (got[p:2 * p], acc.g2_total)):
    assert np.abs(g - r).max() < 2e-3 * np.abs(r).max() + 2e-4, tag
  u_sites, u_acc = vo.step_uniforms(5, np.arange(b), seed, n)
  i_up, i_dn = vo.propose_exchange(cfg, u_sites)
  _, acc_ref, ratios = vo.mc_step(amp, cfg, i_up, i_dn, u_acc)
  mask = eng.mc_step_injected(i_up, i_dn, u_acc)
  band = np.abs(ratios - np.sqrt(u_acc.astype(np.float64))) < 1e-4 * np.maximum(ratios, 1e-30)
  assert np.array_equal(mask[~band], acc_ref[~band]), tag
  close(eng.amplitude()[0], vo.fc_logit(theta, eng.get_configs(), h, L, **kw), 2e-5)
  eng.close()
