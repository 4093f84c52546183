"""GPU parity: libcgsvmc_hip.so (through the C ABI) against the numpy oracle.

Tolerances (fp32 path; the reference computes in float32):
  logits              |d| <= 2e-5 * max(1, |logit|)
  local energies      |d| <= 2e-4 * max(1, |E|)
  gradient sums       |d| <= 2e-3 * ||.||_inf of the vector + 1e-4
  accept masks        bit-exact except where |ratio - sqrt(u)| < 1e-4 * ratio
  proposals / uniforms  bit-exact (integer / same Philox arithmetic)
"""
import numpy as np
import pytest

from oracle import vmc_oracle as vo

pytestmark = pytest.mark.gpu

SHAPES = [
    # n_sites, H, L, B, bonds
    (16, 32, 2, 64, 'chain'),      # BASELINE config 1 (plumbing)
    (16, 32, 2, 64, 'torus4x4'),
    (36, 128, 3, 200, 'torus6x6'),  # config 2 ansatz, ragged batch (not a multiple of 16/128)
    (10, 80, 3, 37, 'chain'),      # reference default fc_layer_size=80 -> padded to 128
    (12, 200, 1, 48, 'chain'),     # single layer, H padded to 256
    (20, 256, 4, 130, 'chain'),    # config-5-like depth
    (14, 160, 3, 70, 'chain'),     # H padded to 192: the 12-tile kernel instantiations
    (150, 256, 6, 20, 'chain'),    # config-5 ansatz; N > 128: general (non-prefetch) sampler path,
                                   # W1 does not fit LDS next to the chain state
]


def _bonds(kind, n):
  if kind == 'chain':
    return vo.chain_bonds(n)
  lx = int(kind[5])
  return vo.torus_bonds(lx, n // lx)


def _make(n, h, L, b, kind, seed=0):
  from cgs_vmc_amd.engine import VmcEngine
  rng = np.random.default_rng(seed)
  theta = vo.init_params(n, h, L, rng)
  theta += (0.05 * rng.standard_normal(theta.size)).astype(np.float32)
  cfg = vo.random_configurations(n, b, np.random.RandomState(seed + 1))
  bonds = _bonds(kind, n)
  eng = VmcEngine(n, b, L, h, seed=2024)
  eng.set_params(theta)
  eng.set_configs(cfg)
  eng.set_bonds(bonds, -1.0, 1.0)
  return eng, theta, cfg, bonds


def _close(a, b, rel, floor=1.0):
  a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
  tol = rel * np.maximum(floor, np.abs(b))
  bad = np.abs(a - b) > tol
  assert not bad.any(), 'max err {} at {} (tol {})'.format(
      np.abs(a - b).max(), np.argmax(np.abs(a - b)), tol[np.argmax(np.abs(a - b))])


def test_mfma_gemm_layouts():
  """A=I style check with asymmetric operands for every stride pattern the gradient uses."""
  from cgs_vmc_amd.engine import VmcEngine
  eng = VmcEngine(4, 16, 1, 32)
  rng = np.random.default_rng(0)
  for (m, n, k) in [(64, 64, 16), (70, 33, 50), (256, 256, 300), (5, 130, 1000), (100, 80, 7)]:
    a = rng.integers(-3, 4, (m, k)).astype(np.float32)
    b = rng.integers(-3, 4, (k, n)).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    np.testing.assert_array_equal(eng.debug_gemm(a, b), ref)            # exact small ints
    np.testing.assert_array_equal(eng.debug_gemm(a.T.copy(), b, trans_a=True), ref)
    np.testing.assert_array_equal(eng.debug_gemm(a, b.T.copy(), trans_b=True), ref)
  eng.close()


@pytest.mark.parametrize('n,h,L,b,kind', SHAPES)
def test_amplitude_matches_oracle(n, h, L, b, kind):
  eng, theta, cfg, _ = _make(n, h, L, b, kind)
  ref = vo.fc_logit(theta, cfg, h, L, dtype=np.float64)
  logit, psi = eng.amplitude(cfg)
  _close(logit, ref, 2e-5)
  logit_c, _ = eng.amplitude()                      # cached path on the engine's chains
  _close(logit_c, ref, 2e-5)
  np.testing.assert_allclose(psi, np.exp(logit.astype(np.float32) + np.float32(10.0)), rtol=1e-6)
  # arbitrary row counts incl. 1 and a non-multiple of 128
  for m in (1, 129):
    c2 = vo.random_configurations(n, m, np.random.RandomState(9))
    _close(eng.amplitude(c2)[0], vo.fc_logit(theta, c2, h, L, dtype=np.float64), 2e-5)
  eng.close()


@pytest.mark.parametrize('n,h,L,b,kind', SHAPES)
def test_local_energy_matches_oracle(n, h, L, b, kind):
  eng, theta, cfg, bonds = _make(n, h, L, b, kind)
  amp = lambda c: vo.fc_psi(theta, c, h, L, dtype=np.float64)
  for jx in (-1.0, 1.0):
    eng.set_bonds(bonds, jx, 1.0)
    diag_ref, off_ref = vo.heisenberg_build(amp, cfg, bonds, jx, 1.0, np.float64)
    ref = diag_ref + off_ref / amp(cfg)
    eloc, mean = eng.local_energy()
    _close(eloc, ref, 2e-4)
    assert abs(mean - ref.mean()) < 2e-4 * max(1.0, abs(ref.mean()))
    diag, off = eng.local_energy_terms()
    _close(diag, diag_ref, 1e-5)
    _close(off, off_ref / amp(cfg), 2e-4)
    # rows evaluated = number of antiparallel bonds (masked rows are never generated)
    n_anti = sum(int(((cfg[:, i] * cfg[:, j]) < 0).sum()) for i, j in bonds)
    assert eng.last_connected_rows() == n_anti
  eng.close()


def test_local_energy_per_bond_couplings():
  """Extension D5: per-bond j_x/j_z reduce to the reference when constant."""
  n, h, L, b = 16, 32, 2, 50
  eng, theta, cfg, bonds = _make(n, h, L, b, 'torus4x4')
  rng = np.random.default_rng(3)
  jx = rng.uniform(-1, 1, len(bonds)).astype(np.float32)
  jz = rng.uniform(0.5, 1.5, len(bonds)).astype(np.float32)
  eng.set_bonds(bonds, jx, jz)
  amp = lambda c: vo.fc_psi(theta, c, h, L, dtype=np.float64)
  ref = vo.local_value(amp, cfg, bonds, jx, jz, dtype=np.float64)
  _close(eng.local_energy()[0], ref, 2e-4)
  eng.close()


def test_constant_wavefunction_closed_form():
  """theta = 0 => psi const => E_loc = 0.25 jz (n_par - n_anti) + 0.5 jx n_anti exactly."""
  n, h, L, b = 16, 32, 2, 64
  eng, theta, cfg, bonds = _make(n, h, L, b, 'torus4x4')
  eng.set_params(np.zeros_like(theta))
  eng.set_bonds(bonds, 0.7, 1.0)
  np.testing.assert_allclose(eng.local_energy()[0],
                             vo.constant_psi_local_energy(cfg, bonds, 0.7, 1.0), rtol=1e-6)
  eng.close()


@pytest.mark.parametrize('n,h,L,b,kind', SHAPES[:4] + SHAPES[-1:])
def test_proposals_bit_exact(n, h, L, b, kind):
  eng, theta, cfg, _ = _make(n, h, L, b, kind)
  for step in (0, 1, 12345678901):
    u_sites, u_acc = vo.step_uniforms(2024, np.arange(b), step, n)
    i_up_ref, i_dn_ref = vo.propose_exchange(cfg, u_sites)
    i_up, i_dn, u = eng.debug_proposals(step)
    np.testing.assert_array_equal(i_up, i_up_ref)
    np.testing.assert_array_equal(i_dn, i_dn_ref)
    np.testing.assert_array_equal(u, u_acc)
  np.testing.assert_array_equal(eng.get_configs(), cfg)   # debug call does not move chains
  eng.close()


def test_chain_offset_keys_rng_by_global_id():
  from cgs_vmc_amd.engine import VmcEngine
  n, h, L, b = 12, 32, 2, 32
  rng = np.random.default_rng(0)
  theta = vo.init_params(n, h, L, rng)
  cfg = vo.random_configurations(n, b, np.random.RandomState(1))
  full = VmcEngine(n, b, L, h, seed=7); full.set_params(theta); full.set_configs(cfg)
  half = VmcEngine(n, b // 2, L, h, seed=7, chain_offset=b // 2)
  half.set_params(theta); half.set_configs(cfg[b // 2:])
  for a, c in zip(full.debug_proposals(5), half.debug_proposals(5)):
    np.testing.assert_array_equal(a[b // 2:], c)
  full.close(); half.close()


@pytest.mark.parametrize('n,h,L,b,kind', SHAPES[:5] + SHAPES[-1:])
def test_injected_mc_step_matches_oracle(n, h, L, b, kind):
  eng, theta, cfg, _ = _make(n, h, L, b, kind)
  amp = lambda c: vo.fc_psi(theta, c, h, L, dtype=np.float64)
  cur = cfg
  for step in range(6):
    u_sites, u_acc = vo.step_uniforms(99, np.arange(b), step, n)
    i_up, i_dn = vo.propose_exchange(cur, u_sites)
    new_ref, acc_ref, ratios = vo.mc_step(amp, cur, i_up, i_dn, u_acc)
    mask = eng.mc_step_injected(i_up, i_dn, u_acc)
    band = np.abs(ratios - np.sqrt(u_acc.astype(np.float64))) < 1e-4 * np.maximum(ratios, 1e-30)
    assert np.array_equal(mask[~band], acc_ref[~band])
    got = eng.get_configs()
    assert (got.sum(1) == cur.sum(1)).all() and (np.abs(got) == 1).all()
    expect = cur.copy()
    rows = np.arange(b)[mask]
    expect[rows, i_dn[mask]] = 1.0
    expect[rows, i_up[mask]] = -1.0
    np.testing.assert_array_equal(got, expect)
    cur = got
    # the cache the kernel writes back is the amplitude of the new chains
    _close(eng.amplitude()[0], vo.fc_logit(theta, cur, h, L, dtype=np.float64), 2e-5)
  eng.close()


@pytest.mark.parametrize('n,h,L,b', [(16, 32, 2, 64),      # prefetched-draw sampler, W1 in LDS
                                     (150, 64, 2, 40),     # 128 < N <= 256: four prefetched site blocks per lane
                                     (256, 64, 2, 24),     # N = 256: every lane's four blocks are sites
                                     (252, 256, 3, 20),    # H = 256 (8 waves), W1 streamed from L2
                                     (300, 64, 2, 24)])    # N > 256: general sampler path
def test_sampler_trajectory_follows_oracle(n, h, L, b):
  """vmc_mc_steps with its own Philox stream reproduces the oracle's chains step by step
  (chains whose accept test falls in the tolerance band are excluded from then on)."""
  eng, theta, cfg, _ = _make(n, h, L, b, 'chain')
  amp = lambda c: vo.fc_psi(theta, c, h, L, dtype=np.float64)
  cur = cfg.copy()
  ok = np.ones(b, bool)
  total_acc = 0
  for step in range(12):
    u_sites, u_acc = vo.step_uniforms(2024, np.arange(b), step, n)
    i_up, i_dn = vo.propose_exchange(cur, u_sites)
    cur, acc, ratios = vo.mc_step(amp, cur, i_up, i_dn, u_acc)
    ok &= ~(np.abs(ratios - np.sqrt(u_acc.astype(np.float64))) < 1e-4 * np.maximum(ratios, 1e-30))
    total_acc += eng.mc_steps(1)
    got = eng.get_configs()
    np.testing.assert_array_equal(got[ok], cur[ok])
  assert ok.sum() > b // 2
  assert eng.step_counter == 12
  assert 0 < total_acc <= 12 * b
  # one launch of 12 steps == 12 launches of 1 step (same counters)
  eng2, _, _, _ = _make(n, h, L, b, 'chain')
  eng2.mc_steps(12)
  np.testing.assert_array_equal(eng2.get_configs()[ok], cur[ok])
  eng.close(); eng2.close()


def test_sweeps_conserve_sz_and_keep_cache_exact():
  n, h, L, b = 36, 128, 3, 100
  eng, theta, cfg, bonds = _make(n, h, L, b, 'torus6x6')
  acc = eng.mc_steps(5 * n)
  assert 0 < acc < 5 * n * b
  got = eng.get_configs()
  assert (np.abs(got) == 1).all() and (got.sum(1) == cfg.sum(1)).all()
  assert (got != cfg).any()
  _close(eng.amplitude()[0], vo.fc_logit(theta, got, h, L, dtype=np.float64), 2e-5)
  amp = lambda c: vo.fc_psi(theta, c, h, L, dtype=np.float64)
  _close(eng.local_energy()[0], vo.local_value(amp, got, bonds, -1.0, 1.0, dtype=np.float64), 2e-4)
  eng.close()


def _oracle_acc(theta, p):
  return vo.Accumulators(p, np.float64)


@pytest.mark.parametrize('n,h,L,b,kind', SHAPES)
def test_energy_gradient_accumulators(n, h, L, b, kind):
  eng, theta, cfg, bonds = _make(n, h, L, b, kind)
  acc = _oracle_acc(theta, theta.size)
  from cgs_vmc_amd import _hip
  eng.reset_accumulators()
  cur = cfg
  for it in range(2):
    vo.energy_gradient_accumulate(acc, theta, cur, bonds, -1.0, 1.0, -10.0, h, L, np.float64)
    eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
    cur = vo.random_configurations(n, b, np.random.RandomState(50 + it))
    eng.set_configs(cur)
  got = eng.get_accumulators()
  p = theta.size
  for name, g, r in (('g1', got[:p], acc.g1_total), ('g2', got[p:2 * p], acc.g2_total)):
    tol = 2e-3 * np.abs(r).max() + 1e-4
    assert np.abs(g - r).max() < tol, (name, np.abs(g - r).max(), tol)
  sc = got[2 * p:]
  assert abs(sc[0] - acc.e_total) < 2e-4 * max(1, abs(acc.e_total))
  assert sc[1] == acc.e_count and sc[4] == acc.g_count
  grad_ref = vo.energy_gradient(acc)
  grad = eng.get_gradient(_hip.VMC_MODE_ENERGY_GRADIENT)
  assert np.abs(grad - grad_ref).max() < 2e-3 * np.abs(grad_ref).max() + 2e-4
  assert abs(eng.mean_energy() - acc.mean_energy()) < 2e-4 * max(1, abs(acc.mean_energy()))
  # TF1 Adam step
  st = vo.AdamState(p)
  th_ref = vo.adam_apply(st, theta, grad, 1e-3, 0.9, 0.99, 1e-8)
  e = eng.apply_adam(_hip.VMC_MODE_ENERGY_GRADIENT, 1e-3, 0.9, 0.99, 1e-8)
  assert abs(e - acc.mean_energy()) < 2e-4 * max(1, abs(acc.mean_energy()))
  np.testing.assert_allclose(eng.get_params(), th_ref, rtol=0, atol=2e-6)
  # parameters changed => amplitudes follow the new parameters
  _close(eng.amplitude()[0], vo.fc_logit(eng.get_params(), cur, h, L, dtype=np.float64), 2e-5)
  eng.close()


@pytest.mark.parametrize('n,h,L,b,kind', SHAPES[:4])
def test_log_overlap_itswo_accumulators(n, h, L, b, kind):
  from cgs_vmc_amd import _hip
  eng, theta, cfg, bonds = _make(n, h, L, b, kind)
  eng.transfer_params()                               # omega <- psi
  theta_w = theta.copy()
  rng = np.random.default_rng(8)
  theta2 = theta + (0.02 * rng.standard_normal(theta.size)).astype(np.float32)
  eng.set_params(theta2)                              # psi moved on, omega frozen
  eng.set_shift(-9.0)                                 # psi's shift was updated, omega's stays -10
  acc = _oracle_acc(theta, theta.size)
  vo.log_overlap_accumulate(acc, theta2, theta_w, cfg, bonds, -1.0, 1.0, -9.0, -10.0, 0.12, h, L,
                            np.float64)
  eng.reset_accumulators()
  eng.accumulate(_hip.VMC_MODE_LOG_OVERLAP_ITSWO, 0.12)
  got = eng.get_accumulators()
  p = theta.size
  for name, g, r in (('g1', got[:p], acc.g1_total), ('g2', got[p:2 * p], acc.g2_total)):
    tol = 2e-3 * np.abs(r).max() + 1e-4
    assert np.abs(g - r).max() < tol, (name, np.abs(g - r).max(), tol)
  sc = got[2 * p:]
  assert abs(sc[0] - acc.e_total) < 2e-4 * max(1, abs(acc.e_total))
  assert abs(sc[2] - acc.r_total) < 2e-4 * max(1, abs(acc.r_total))
  grad_ref = vo.log_overlap_gradient(acc)
  grad = eng.get_gradient(_hip.VMC_MODE_LOG_OVERLAP_ITSWO)
  assert np.abs(grad - grad_ref).max() < 2e-3 * np.abs(grad_ref).max() + 2e-4
  eng.close()


def test_update_norm_rule():
  n, h, L, b = 16, 32, 2, 64
  eng, theta, cfg, _ = _make(n, h, L, b, 'chain')
  logit = vo.fc_logit(theta, cfg, h, L, dtype=np.float64)
  eng.update_norm(1e10)                    # psi ~ e^{10+logit} << 1e10: unchanged (+= 0)
  assert eng.get_shift() == -10.0
  eng.set_shift(-40.0)                     # psi ~ e^{40+logit} > 1e10: shift grows by the excess
  eng.update_norm(1e10)
  expect = -40.0 + (logit.max() + 40.0 - np.log(1e10))
  assert abs(eng.get_shift() - expect) < 1e-4
  _, psi = eng.amplitude()
  assert abs(psi.max() - 1e10) < 1e-3 * 1e10
  eng.close()


def test_error_behaviour():
  from cgs_vmc_amd.engine import VmcEngine
  with pytest.raises(ValueError):
    VmcEngine(16, 8, 2, 32, nonlinearity='swish')         # not in layers.NONLINEARITIES
  with pytest.raises(ValueError):
    VmcEngine(16, 0, 2, 32)
  eng = VmcEngine(16, 8, 2, 32)
  with pytest.raises(ValueError):
    eng.set_configs(np.ones((4, 16), np.float32))        # graph_builders.py:117-118
  with pytest.raises(ValueError):
    eng.set_configs(np.zeros((8, 16), np.float32))
  with pytest.raises(ValueError):
    eng.set_bonds([(0, 16)], 1.0, 1.0)
  with pytest.raises(Exception):
    eng.local_energy()                                     # no params / bonds yet
  eng.close()


@pytest.mark.parametrize('n,h,L,b,bonds', [
    (9, 32, 2, 1, [(0, 1)]),                      # one chain, odd site count (4 down spins), one bond
    (2, 16, 1, 3, [(0, 1)]),                      # smallest lattice the kernels accept
    (33, 64, 2, 17, [(i, (i + 2) % 33) for i in range(33)]),   # N not a multiple of 4, odd batch
])
def test_edge_shapes(n, h, L, b, bonds):
  """Ragged / minimal shapes through every kernel: amplitudes, local energy, proposals, a few
  sweeps, gradient accumulators."""
  from cgs_vmc_amd import _hip
  from cgs_vmc_amd.engine import VmcEngine
  rng = np.random.default_rng(4)
  theta = vo.init_params(n, h, L, rng)
  theta += (0.05 * rng.standard_normal(theta.size)).astype(np.float32)
  cfg = vo.random_configurations(n, b, np.random.RandomState(2))
  eng = VmcEngine(n, b, L, h, seed=11)
  eng.set_params(theta); eng.set_configs(cfg); eng.set_bonds(bonds, -1.0, 1.0)
  amp = lambda c: vo.fc_psi(theta, c, h, L, dtype=np.float64)
  _close(eng.amplitude()[0], vo.fc_logit(theta, cfg, h, L, dtype=np.float64), 2e-5)
  _close(eng.local_energy()[0], vo.local_value(amp, cfg, bonds, -1.0, 1.0, dtype=np.float64), 2e-4)
  u_sites, u_acc = vo.step_uniforms(11, np.arange(b), 0, n)
  i_up_ref, i_dn_ref = vo.propose_exchange(cfg, u_sites)
  i_up, i_dn, u = eng.debug_proposals(0)
  np.testing.assert_array_equal(i_up, i_up_ref); np.testing.assert_array_equal(i_dn, i_dn_ref)
  np.testing.assert_array_equal(u, u_acc)
  eng.mc_steps(3 * n)
  got = eng.get_configs()
  assert (np.abs(got) == 1).all() and (got.sum(1) == cfg.sum(1)).all()
  _close(eng.amplitude()[0], vo.fc_logit(theta, got, h, L, dtype=np.float64), 2e-5)
  acc = vo.Accumulators(theta.size, np.float64)
  vo.energy_gradient_accumulate(acc, theta, got, bonds, -1.0, 1.0, -10.0, h, L, np.float64)
  eng.reset_accumulators(); eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  a = eng.get_accumulators(); p = theta.size
  for g, r in ((a[:p], acc.g1_total), (a[p:2 * p], acc.g2_total)):
    assert np.abs(g - r).max() < 2e-3 * np.abs(r).max() + 1e-4
  eng.close()


def test_lattice_size_limit_is_checked_at_create_and_large_lattices_run():
  """The sampler keeps 16 chains' state in LDS: shapes beyond 160 KiB are refused at vmc_create
  (NotImplementedError), the largest that fit run and match the oracle."""
  from cgs_vmc_amd.engine import VmcEngine
  with pytest.raises(NotImplementedError, match='LDS'):
    VmcEngine(2048, 16, 3, 256)
  n, h, L, b = 1400, 256, 3, 16              # 159 KiB of chain state
  rng = np.random.default_rng(0)
  theta = vo.init_params(n, h, L, rng)
  cfg = vo.random_configurations(n, b, np.random.RandomState(1))
  eng = VmcEngine(n, b, L, h, seed=3)
  eng.set_params(theta); eng.set_configs(cfg)
  _close(eng.amplitude()[0], vo.fc_logit(theta, cfg, h, L, dtype=np.float64), 2e-5)
  acc = eng.mc_steps(20)
  got = eng.get_configs()
  assert 0 <= acc <= 20 * b and (np.abs(got) == 1).all() and (got.sum(1) == cfg.sum(1)).all()
  _close(eng.amplitude()[0], vo.fc_logit(theta, got, h, L, dtype=np.float64), 2e-5)
  eng.close()


@pytest.mark.parametrize('n,h,L', [(100, 256, 3), (36, 256, 2), (36, 128, 3), (16, 32, 2), (150, 256, 6)])
def test_uniform_chain_is_safe_next_to_normal_chains(n, h, L):
  """vmc_set_configs accepts any +-1 rows, also all-up / all-down ones, for which the
  exchange move does not exist (graph_builders.py:62-71 would write a spin of +-3).  The
  sortable-key sampler used to turn `no key` into site 255: W1 rows past the allocation and a
  neighbouring chain's spins in LDS.  Now such a chain proposes the null move with a NaN
  acceptance uniform (frozen); the other sampler variants take argmin / argmax over all sites like
  the reference and stay inside the lattice.  Either way the NEIGHBOURS must walk exactly the
  chains they walk without the uniform rows (chains are independent)."""
  from cgs_vmc_amd.engine import VmcEngine
  b = 48
  rng = np.random.default_rng(n + h)
  theta = vo.init_params(n, h, L, rng)
  cfg = vo.random_configurations(n, b, np.random.RandomState(2))
  bad = cfg.copy()
  bad[5] = 1.0; bad[17] = -1.0; bad[31] = 1.0; bad[47] = -1.0
  normal = np.array([i for i in range(b) if i not in (5, 17, 31, 47)])
  out = []
  for start in (cfg, bad):
    eng = VmcEngine(n, b, L, h, seed=99)
    eng.set_params(theta); eng.set_configs(start)
    eng.mc_steps(3 * n)
    out.append(eng.get_configs())
    eng.close()
  np.testing.assert_array_equal(out[1][normal], out[0][normal])
  assert (np.abs(out[1]) == 1.0).all()
  if h == 256 and n <= 128:          # the production (key hand-over) sampler: frozen
    np.testing.assert_array_equal(out[1][[5, 17, 31, 47]], bad[[5, 17, 31, 47]])


@pytest.mark.parametrize('n,h,L,b,kind', [(16, 32, 2, 64, 'torus4x4'), (36, 128, 3, 200, 'torus6x6')])
def test_log_domain_stays_finite_where_linear_psi_overflows(n, h, L, b, kind):
  """SURVEY B9 / BASELINE.md "overflow semantics".  The reference works with psi itself:
  psi = exp(logit - shift) (wavefunctions.py:232), ratio = psi'/psi (graph_builders.py:75),
  E_loc = diag + offdiag / psi (operators.py:259).  With exp_norm_shift = -100 every psi of this
  network overflows float32: the fp32 twin of the oracle -- the reference's arithmetic -- gets
  inf/inf = NaN for E_loc and for the acceptance ratio (no move is ever accepted).  The HIP path
  keeps logits and differences of logits, in which the shift cancels: its local energies, accept
  masks and gradient sums stay finite and equal the fp64 twin (which does not overflow) within the
  usual tolerances; vmc_amplitude still REPORTS psi = inf, as the reference's tensor would."""
  eng, theta, cfg, bonds = _make(n, h, L, b, kind)
  shift = -100.0
  eng.set_shift(shift)
  amp32 = lambda c: vo.fc_psi(theta, c, h, L, shift, dtype=np.float32)
  amp64 = lambda c: vo.fc_psi(theta, c, h, L, shift, dtype=np.float64)
  with np.errstate(over='ignore', invalid='ignore'):
    assert np.isinf(amp32(cfg)).all()
    e32 = vo.local_value(amp32, cfg, bonds, -1.0, 1.0, dtype=np.float32)
  assert np.isnan(e32).all()                                    # the reference's fp32 result
  e64 = vo.local_value(amp64, cfg, bonds, -1.0, 1.0, dtype=np.float64)
  eloc, mean = eng.local_energy()
  assert np.isfinite(eloc).all() and np.isfinite(mean)
  _close(eloc, e64, 2e-4)
  logit, psi = eng.amplitude()
  assert np.isfinite(logit).all() and np.isinf(psi).all()
  # Metropolis accept: NaN > sqrt(u) is False in the fp32 twin, the log-domain test follows fp64
  u_sites, u_acc = vo.step_uniforms(99, np.arange(b), 0, n)
  i_up, i_dn = vo.propose_exchange(cfg, u_sites)
  with np.errstate(over='ignore', invalid='ignore'):
    _, acc32, _ = vo.mc_step(amp32, cfg, i_up, i_dn, u_acc)
  _, acc64, ratios = vo.mc_step(amp64, cfg, i_up, i_dn, u_acc)
  assert not acc32.any() and acc64.any()
  mask = eng.mc_step_injected(i_up, i_dn, u_acc)
  band = np.abs(ratios - np.sqrt(u_acc.astype(np.float64))) < 1e-4 * np.maximum(ratios, 1e-30)
  assert np.array_equal(mask[~band], acc64[~band])
  # gradient accumulators: finite, equal to the fp64 twin
  eng.set_configs(cfg)
  eng.reset_accumulators()
  eng.accumulate(0)
  acc = vo.Accumulators(theta.size, np.float64)
  vo.energy_gradient_accumulate(acc, theta, cfg, bonds, -1.0, 1.0, shift, h, L, np.float64)
  got = eng.get_accumulators()
  assert np.isfinite(got).all()
  p = theta.size
  for lo, ref in ((0, acc.g1_total), (p, acc.g2_total)):
    assert np.abs(got[lo:lo + p] - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-4
  # update_norm moves the shift by log(max psi) - log(1e10) taken in the logit domain
  eng.update_norm(1e10)
  expect = shift + (float(vo.fc_logit(theta, cfg, h, L, dtype=np.float64).max()) - shift - np.log(1e10))
  assert abs(eng.get_shift() - expect) < 1e-4 * abs(expect)
  eng.close()


@pytest.mark.parametrize('ansatz,h', [('fully_connected', 64), ('rbm', 48), ('fully_connected', 320)])
def test_bond_census_follows_every_writer_of_the_chains(ansatz, h):
  """The dense sampler leaves the bond census (antiparallel counts, diagonal terms) of its final chains
  (sweep16.hpp epilogue); every other writer of the chains or the bonds must invalidate it.  After each kind
  of change the local energies are those of the oracle on the chains the engine reports, and the row count is
  the number of antiparallel bonds."""
  from cgs_vmc_amd.engine import VmcEngine
  n, L, b = 16, 2, 37
  rbm = ansatz == 'rbm'
  rng = np.random.default_rng(5)
  theta = (vo.rbm_init_params if rbm else vo.init_params)(n, h, L, rng)
  psi = (lambda c: vo.rbm_psi(theta, c, h, L, dtype=np.float64)) if rbm else \
        (lambda c: vo.fc_psi(theta, c, h, L, dtype=np.float64))
  eng = VmcEngine(n, b, L, h, seed=9, ansatz=ansatz)
  eng.set_params(theta)
  chain, torus = vo.chain_bonds(n), vo.torus_bonds(4, 4)
  state = {'bonds': chain}

  def check():
    c = eng.get_configs()
    e = eng.local_energy()[0]
    ref = vo.local_value(psi, c, state['bonds'], -1.0, 1.0, dtype=np.float64)
    assert np.abs(e - ref).max() < 2e-4 * max(1.0, np.abs(ref).max())
    assert eng.last_connected_rows() == sum(int((c[:, i] * c[:, j] < 0).sum()) for (i, j) in state['bonds'])

  eng.set_bonds(chain, -1.0, 1.0)
  eng.set_configs(vo.random_configurations(n, b, np.random.RandomState(1)))
  check()                                   # census by its own launch
  eng.mc_steps(7); check()                  # census left by the sampler
  eng.mc_steps(3); eng.mc_steps(2); check() # two launches in a row
  u_sites, u_acc = vo.step_uniforms(3, np.arange(b), 0, n)
  i_up, i_dn = vo.propose_exchange(eng.get_configs(), u_sites)
  eng.mc_step_injected(i_up, i_dn, u_acc); check()          # injected step: no census
  eng.mc_steps(5)
  eng.set_bonds(torus, -1.0, 1.0); state['bonds'] = torus; check()   # new bonds after a sampler launch
  eng.mc_steps(4)
  eng.set_configs(vo.random_configurations(n, b, np.random.RandomState(2))); check()   # new chains after one
  eng.reset_accumulators(); eng.accumulate(0); eng.mc_steps(n); eng.accumulate(0); check()   # the training order
  eng.close()
