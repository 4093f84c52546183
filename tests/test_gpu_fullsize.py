"""GPU: BASELINE.json's full-size configurations through size-independent properties
(the oracle is too slow there, so it is only used on a few sampled chains)."""
import numpy as np
import pytest

from oracle import vmc_oracle as vo

pytestmark = pytest.mark.gpu

# name: (lx, ly, next-nearest bonds too, L, H, chains per GPU)
FULL = {
    'config2_6x6_h128_b1024': (6, 6, False, 3, 128, 1024),
    'config3_10x10_h256_b4096': (10, 10, False, 3, 256, 4096),
    # BASELINE config 5, one GPU's shard: NN + NNN bonds with per-bond couplings J1 = 1, J2 = 0.5
    'config5_16x16_j1j2_h256_L6_b1024': (16, 16, True, 6, 256, 1024),
    # bench.py's 512-unit workload: k_sweep16<32> / k_tail_lds<32>
    'wide_10x10_h512_b4096': (10, 10, False, 3, 512, 4096),
}


def _couplings(name, bonds, jx_sign=-1.0):
  """(j_x, j_z) of the workload: scalars for the NN lattices, per-bond arrays for J1-J2."""
  if not FULL[name][2]:
    return jx_sign, 1.0
  nb = len(bonds)
  j = np.concatenate([np.ones(nb // 2), 0.5 * np.ones(nb // 2)]).astype(np.float32)
  return jx_sign * j, j


def _setup(name, seed=2024, chains=None, offset=0):
  import bench
  from cgs_vmc_amd.engine import VmcEngine
  lx, ly, nnn, L, h, b = FULL[name]
  n = lx * ly
  theta, cfg = bench.make_inputs(n, h, L, b, 0)
  bonds = vo.torus_bonds(lx, ly, nnn)
  if nnn:
    assert len(bonds) == 4 * n
  if chains is not None:
    cfg = cfg[offset:offset + chains]
    b = chains
  eng = VmcEngine(n, b, L, h, seed=seed, chain_offset=offset)
  eng.set_params(theta)
  eng.set_configs(cfg)
  eng.set_bonds(bonds, *_couplings(name, bonds))
  return eng, theta, cfg, bonds, (n, h, L, b)


@pytest.mark.parametrize('name', sorted(FULL))
def test_sweeps_conserve_sz_are_deterministic_and_shard_invariant(name):
  eng, theta, cfg, bonds, (n, h, L, b) = _setup(name)
  acc = eng.mc_steps(2 * n)
  out = eng.get_configs()
  assert (np.abs(out) == 1).all() and (out.sum(1) == cfg.sum(1)).all()
  # (the deep random-init network of config 5 is nearly flat: almost every move is accepted)
  assert 0.02 < acc / (2.0 * n * b) < (0.98 if L <= 3 else 1.0) and (out != cfg).any()
  # same seed, same inputs -> bit-identical chains, logits and energies
  eng2, *_ = _setup(name)
  eng2.mc_steps(2 * n)
  np.testing.assert_array_equal(eng2.get_configs(), out)
  np.testing.assert_array_equal(eng2.amplitude()[0], eng.amplitude()[0])
  np.testing.assert_array_equal(eng2.local_energy()[0], eng.local_energy()[0])
  # a shard (second half of the chains, RNG keyed by global chain id) walks the same chains
  half, *_ = _setup(name, chains=b // 2, offset=b // 2)
  half.mc_steps(2 * n)
  np.testing.assert_array_equal(half.get_configs(), out[b // 2:])
  # cache written back by the sweep kernel == amplitudes recomputed from the chains
  logit_cached = eng.amplitude()[0]
  logit_fresh = eng.amplitude(out)[0]
  assert np.abs(logit_cached - logit_fresh).max() < 2e-5 * max(1.0, np.abs(logit_fresh).max())
  # spot-check against the oracle (fp64) on 48 chains
  idx = np.random.default_rng(0).choice(b, 48, replace=False)
  amp = lambda c: vo.fc_psi(theta, c, h, L, dtype=np.float64)
  ref = vo.fc_logit(theta, out[idx], h, L, dtype=np.float64)
  assert np.abs(logit_cached[idx] - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
  e_ref = vo.local_value(amp, out[idx], bonds, *_couplings(name, bonds), dtype=np.float64)
  e = eng.local_energy()[0]
  assert np.abs(e[idx] - e_ref).max() < 2e-4 * max(1.0, np.abs(e_ref).max())
  for x in (eng, eng2, half):
    x.close()


@pytest.mark.parametrize('name', sorted(FULL))
def test_constant_wavefunction_closed_form_at_full_size(name):
  eng, theta, cfg, bonds, (n, h, L, b) = _setup(name)
  eng.set_params(np.zeros_like(theta))
  jx, jz = _couplings(name, bonds, jx_sign=0.6)
  eng.set_bonds(bonds, jx, jz)
  np.testing.assert_allclose(eng.local_energy()[0],
                             vo.constant_psi_local_energy(cfg, bonds, jx, jz), rtol=1e-6, atol=1e-5)
  eng.close()


@pytest.mark.parametrize('name', sorted(FULL))
def test_accumulators_are_additive_over_calls_and_shards(name):
  from cgs_vmc_amd import _hip
  eng, theta, cfg, bonds, (n, h, L, b) = _setup(name)
  p = theta.size
  eng.reset_accumulators()
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  one = eng.get_accumulators().astype(np.float64)
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)          # same chains again: totals double
  two = eng.get_accumulators().astype(np.float64)
  np.testing.assert_allclose(two[:2 * p], 2 * one[:2 * p], rtol=1e-6, atol=1e-6 * np.abs(one).max())
  assert two[2 * p + 1] == 2 * b and two[2 * p + 4] == 2
  # shards: sum of the two halves' accumulators == whole batch (fp32 re-association only)
  parts = []
  for off in (0, b // 2):
    sh, *_ = _setup(name, chains=b // 2, offset=off)
    sh.reset_accumulators()
    sh.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
    parts.append(sh.get_accumulators().astype(np.float64))
    sh.close()
  tot = parts[0] + parts[1]
  scale = np.abs(one[:2 * p]).max()
  assert np.abs(tot[:2 * p] - one[:2 * p]).max() < 1e-4 * scale
  assert abs(tot[2 * p] - one[2 * p]) < 1e-4 * abs(one[2 * p]) and tot[2 * p + 1] == b
  # gradient against the oracle on a 64-chain subset engine (same code path, smaller batch)
  sub, *_ = _setup(name, chains=64, offset=0)
  sub.reset_accumulators()
  sub.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  acc = vo.Accumulators(p, np.float64)
  jx, jz = _couplings(name, bonds)
  vo.energy_gradient_accumulate(acc, theta, cfg[:64], bonds, jx, jz, -10.0, h, L, np.float64)
  g = sub.get_gradient(_hip.VMC_MODE_ENERGY_GRADIENT)
  gref = vo.energy_gradient(acc)
  assert np.abs(g - gref).max() < 2e-3 * np.abs(gref).max() + 2e-4
  sub.close(); eng.close()


@pytest.mark.parametrize('name', sorted(FULL))
def test_full_size_engine_matches_oracle_on_sampled_chains(name):
  """The full-size engine (all chains resident, production kernels and tile counts) against the
  fp64 oracle on every 64th chain: logits, local energies, and -- after a sweep of the whole
  batch -- the same quantities on the moved chains."""
  eng, theta, cfg, bonds, (n, h, L, b) = _setup(name)
  pick = np.arange(0, b, 64)
  amp = lambda c: vo.fc_psi(theta, c, h, L, dtype=np.float64)

  def check(configs):
    sub = configs[pick]
    ref_logit = vo.fc_logit(theta, sub, h, L, dtype=np.float64)
    ref_eloc = vo.local_value(amp, sub, bonds, *_couplings(name, bonds), dtype=np.float64)
    logit = eng.amplitude()[0][pick]
    eloc = eng.local_energy()[0][pick]
    assert np.abs(logit - ref_logit).max() <= 2e-5 * max(1.0, np.abs(ref_logit).max())
    assert np.abs(eloc - ref_eloc).max() <= 2e-4 * max(1.0, np.abs(ref_eloc).max())

  check(cfg)
  eng.mc_steps(n)
  moved = eng.get_configs()
  assert (moved != cfg).any()
  check(moved)
  eng.close()


# the convolutional bench workloads (bench.py WORKLOADS) at their full sizes
FULL_CONV = {
    'conv10x10_5x16k5_b4096': ('conv_2d', 10, 10, False, 5, 16, 5, 4096),
    'resnet10x10_2x16k5_b4096': ('res_net_2d', 10, 10, False, 2, 16, 5, 4096),
    'conv16x16j1j2_5x16k5_b1024': ('conv_2d', 16, 16, True, 5, 16, 5, 1024),
    'conv10x10_3x32k3_b4096': ('conv_2d', 10, 10, False, 3, 32, 3, 4096),      # two channel blocks
}


@pytest.mark.parametrize('name', sorted(FULL_CONV))
def test_conv_workloads_full_size_properties(name):
  """Sz conservation, bit reproducibility, shard invariance, exact logit cache and oracle spot checks
  (logits, local energies) on sampled chains of the full-size convolutional workloads."""
  import bench
  from cgs_vmc_amd.engine import VmcEngine
  ansatz, lx, ly, nnn, L, f, k, b = FULL_CONV[name]
  n = lx * ly
  geom = (f, k, ly, lx)                  # site = x + lx * y: size_x = ly, size_y = lx (bench.py)
  theta, cfg = bench.make_inputs(n, f, L, b, 0, ansatz, k)
  bonds = vo.torus_bonds(lx, ly, nnn)
  if nnn:
    jz = np.concatenate([np.ones(len(bonds) // 2), 0.5 * np.ones(len(bonds) // 2)]).astype(np.float32)
    jx = -jz
  else:
    jx, jz = -1.0, 1.0

  def make(chains=None, offset=0):
    c = cfg if chains is None else cfg[offset:offset + chains]
    eng = VmcEngine(n, len(c), L, f, seed=2024, chain_offset=offset, ansatz=ansatz, kernel_size=k,
                    size_x=ly, size_y=lx)
    eng.set_params(theta); eng.set_configs(c); eng.set_bonds(bonds, jx, jz)
    return eng

  steps = n // 2
  eng = make()
  acc = eng.mc_steps(steps)
  out = eng.get_configs()
  assert (np.abs(out) == 1).all() and (out.sum(1) == cfg.sum(1)).all() and (out != cfg).any()
  assert 0 < acc <= steps * b
  eng2 = make()
  eng2.mc_steps(steps)
  np.testing.assert_array_equal(eng2.get_configs(), out)
  np.testing.assert_array_equal(eng2.local_energy()[0], eng.local_energy()[0])
  half = make(chains=b // 4, offset=b // 2)
  half.mc_steps(steps)
  np.testing.assert_array_equal(half.get_configs(), out[b // 2:b // 2 + b // 4])
  idx = np.random.default_rng(0).choice(b, 12, replace=False)
  logit_cached = eng.amplitude()[0]
  ref, scale = vo.conv_forward(theta, out[idx], ansatz, geom, L, 'relu', np.float64, return_tape='scale')
  assert (np.abs(logit_cached[idx] - ref) <= 1e-6 * scale + 2e-5).all()
  assert (np.abs(eng.amplitude(out[idx])[0] - ref) <= 1e-6 * scale + 2e-5).all()
  psi_fn = vo.ANSATZ[ansatz][0]
  amp = lambda c: psi_fn(theta, c, geom, L, nonlinearity='relu', dtype=np.float64)
  e_ref = vo.local_value(amp, out[idx], bonds, jx, jz, dtype=np.float64)
  e = eng.local_energy()[0]
  assert np.abs(e[idx] - e_ref).max() < 2e-4 * max(1.0, np.abs(e_ref).max())
  for x in (eng, eng2, half):
    x.close()
