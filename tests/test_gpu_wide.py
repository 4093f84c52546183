"""GPU parity of fully_connected with fc_layer_size > 256 (utils.py:105 allows any size) against the
numpy oracle, through the C ABI: networks of at most 512 units (any hidden activation) run
the fused kernels padded to 384 or 512 units (k_sweep16<24|32>, k_tail_lds / k_tail0,
k_backprop16<24|32>; kernel_path() == 1), everything beyond 512 units the general path of
csrc/wide.hip (materialised rows + the library's fp32-MFMA GEMM, a few launches per mc_step;
kernel_path() == 2).  Tolerances as tests/test_gpu_engine.py:
logits 2e-5 * max(1, |logit|), local energies 2e-4 * max(1, |E|), gradient sums
2e-3 * ||.||_inf + 1e-4, accept masks bit-exact outside |ratio - sqrt(u)| < 1e-4 ratio, proposals
bit-exact."""
import numpy as np
import pytest

from oracle import vmc_oracle as vo

pytestmark = pytest.mark.gpu

WIDE_SHAPES = [
    # n_sites, H, num_layers, B, bonds, nonlinearity
    (16, 320, 2, 40, 'torus4x4', 'relu'),
    (36, 512, 3, 64, 'torus6x6', 'relu'),
    (12, 300, 1, 23, 'chain', 'tanh'),        # no H x H layer (k_tail0), H not a multiple of 64
    (20, 264, 2, 17, 'chain', 'sigmoid'),     # non-relu at 384 padded units: general sampler variant
    (10, 384, 3, 50, 'chain', 'relu'),        # exactly 24 unit tiles, two H x H layers
    (24, 500, 2, 33, 'chain', 'relu'),        # padded to 512, ragged batch
    (150, 272, 4, 21, 'chain', 'relu'),       # N > 128: the general (non-prefetch) sampler at 384 units
    (16, 400, 1, 30, 'chain', 'relu'),        # no H x H layer
    (36, 448, 3, 48, 'torus6x6', 'tanh'),     # non-relu at 512 padded units, two H x H layers
    (16, 300, 2, 25, 'torus4x4', 'identity'),
    (12, 512, 2, 31, 'chain', 'sigmoid'),
    (16, 320, 2, 27, 'torus4x4', 'cos'),       # cos: f'(z) arrays next to the activations, 384 padded units
    (20, 512, 3, 22, 'chain', 'cos'),
    (16, 640, 2, 19, 'chain', 'relu'),        # more than 512 units: general path
    (16, 640, 3, 21, 'torus4x4', 'cos'),      # ... with the cosine (round 4): back-propagation through the stored f'(z)
    (12, 1000, 2, 14, 'chain', 'tanh'),
]


def _expected_path(h):
  return 1 if h <= 512 else 2


def _bonds(kind, n):
  if kind == 'chain':
    return vo.chain_bonds(n)
  lx = int(kind[5:].split('x')[0])
  return vo.torus_bonds(lx, n // lx)


def _make(n, h, L, b, kind, nonlin, seed=0):
  from cgs_vmc_amd.engine import VmcEngine
  rng = np.random.default_rng(seed)
  theta = vo.init_params(n, h, L, rng)
  theta += (0.02 * rng.standard_normal(theta.size)).astype(np.float32)
  cfg = vo.random_configurations(n, b, np.random.RandomState(seed + 1))
  bonds = _bonds(kind, n)
  eng = VmcEngine(n, b, L, h, nonlinearity=nonlin, seed=2024)
  assert eng.num_params == theta.size
  assert eng.kernel_path() == _expected_path(h)        # 257..512 units run fused, beyond: general path
  eng.set_params(theta); eng.set_configs(cfg); eng.set_bonds(bonds, -1.0, 1.0)
  return eng, theta, cfg, bonds


def _close(a, b, rel, floor=1.0):
  a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
  tol = rel * np.maximum(floor, np.abs(b))
  assert (np.abs(a - b) <= tol).all(), 'max err {} (tol {})'.format(np.abs(a - b).max(), tol.min())


@pytest.mark.parametrize('n,h,L,b,kind,nonlin', WIDE_SHAPES)
def test_wide_amplitude_local_energy_and_sampler(n, h, L, b, kind, nonlin):
  eng, theta, cfg, bonds = _make(n, h, L, b, kind, nonlin)
  logit_ref = lambda c: vo.fc_logit(theta, c, h, L, nonlinearity=nonlin, dtype=np.float64)
  amp = lambda c: vo.fc_psi(theta, c, h, L, nonlinearity=nonlin, dtype=np.float64)
  _close(eng.amplitude(cfg)[0], logit_ref(cfg), 2e-5)
  _close(eng.amplitude()[0], logit_ref(cfg), 2e-5)
  for jx in (-1.0, 1.0):
    eng.set_bonds(bonds, jx, 1.0)
    _close(eng.local_energy()[0], vo.local_value(amp, cfg, bonds, jx, 1.0, dtype=np.float64), 2e-4)
  u_sites, u_acc = vo.step_uniforms(2024, np.arange(b), 5, n)
  i_up, i_dn = vo.propose_exchange(cfg, u_sites)
  g_up, g_dn, g_u = eng.debug_proposals(5)
  np.testing.assert_array_equal(g_up, i_up); np.testing.assert_array_equal(g_dn, i_dn)
  np.testing.assert_array_equal(g_u, u_acc)
  cur = cfg
  for step in range(3):
    u_sites, u_acc = vo.step_uniforms(99, np.arange(b), step, n)
    i_up, i_dn = vo.propose_exchange(cur, u_sites)
    _, acc_ref, ratios = vo.mc_step(amp, cur, i_up, i_dn, u_acc)
    mask = eng.mc_step_injected(i_up, i_dn, u_acc)
    band = np.abs(ratios - np.sqrt(u_acc.astype(np.float64))) < 1e-4 * np.maximum(ratios, 1e-30)
    assert np.array_equal(mask[~band], acc_ref[~band])
    expect = cur.copy()
    rows = np.arange(b)[mask]
    expect[rows, i_dn[mask]] = 1.0
    expect[rows, i_up[mask]] = -1.0
    got = eng.get_configs()
    np.testing.assert_array_equal(got, expect)
    cur = got
    _close(eng.amplitude()[0], logit_ref(cur), 2e-5)
  eng.step_counter = 0
  ok = np.ones(b, bool)
  ref = cur.copy()
  for step in range(8):
    u_sites, u_acc = vo.step_uniforms(2024, np.arange(b), step, n)
    i_up, i_dn = vo.propose_exchange(ref, u_sites)
    ref, acc, ratios = vo.mc_step(amp, ref, i_up, i_dn, u_acc)
    ok &= ~(np.abs(ratios - np.sqrt(u_acc.astype(np.float64))) < 1e-4 * np.maximum(ratios, 1e-30))
  accepted = eng.mc_steps(8)
  got = eng.get_configs()
  np.testing.assert_array_equal(got[ok], ref[ok])
  assert ok.sum() > b // 2 and 0 <= accepted <= 8 * b
  _close(eng.local_energy()[0], vo.local_value(amp, got, bonds, 1.0, 1.0, dtype=np.float64), 2e-4)
  eng.close()


@pytest.mark.parametrize('n,h,L,b,kind,nonlin', WIDE_SHAPES)
def test_wide_energy_gradient_accumulators(n, h, L, b, kind, nonlin):
  from cgs_vmc_amd import _hip
  eng, theta, cfg, bonds = _make(n, h, L, b, kind, nonlin)
  acc = vo.Accumulators(theta.size, np.float64)
  eng.reset_accumulators()
  vo.energy_gradient_accumulate(acc, theta, cfg, bonds, -1.0, 1.0, -10.0, h, L, np.float64,
                                nonlinearity=nonlin)
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  eng.mc_steps(3)
  cur = eng.get_configs()
  vo.energy_gradient_accumulate(acc, theta, cur, bonds, -1.0, 1.0, -10.0, h, L, np.float64,
                                nonlinearity=nonlin)
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  got = eng.get_accumulators()
  p = theta.size
  for name, g, r in (('g1', got[:p], acc.g1_total), ('g2', got[p:2 * p], acc.g2_total)):
    tol = 2e-3 * np.abs(r).max() + 1e-4
    assert np.abs(g - r).max() < tol, (name, np.abs(g - r).max(), tol)
  grad_ref = vo.energy_gradient(acc)
  grad = eng.get_gradient(_hip.VMC_MODE_ENERGY_GRADIENT)
  assert np.abs(grad - grad_ref).max() < 2e-3 * np.abs(grad_ref).max() + 2e-4
  eng.apply_adam(_hip.VMC_MODE_ENERGY_GRADIENT, 1e-3, 0.9, 0.99, 1e-8)
  th = eng.get_params()
  _close(eng.amplitude()[0], vo.fc_logit(th, cur, h, L, nonlinearity=nonlin, dtype=np.float64), 2e-5)
  eng.close()


def test_wide_limits():
  from cgs_vmc_amd.engine import VmcEngine
  with pytest.raises(NotImplementedError):
    VmcEngine(16, 8, 2, 4100, ansatz='rbm')
  for ansatz in ('fully_connected', 'rbm'):
    eng = VmcEngine(16, 8, 2, 640, ansatz=ansatz)     # beyond 512 units: general path (SR: tests/test_gpu_sr.py)
    assert eng.kernel_path() == 2
    eng.sr_reserve(2)
    eng.sr_reserve(0)
    eng.close()


def test_wide_fast_and_general_path_agree(monkeypatch):
  """CGS_VMC_WIDE_FAST=0 (general path) and the fused kernels give the same logits, local energies and
  accumulators for one relu network within the stated fp32 tolerances."""
  from cgs_vmc_amd import _hip
  outs = []
  for fast in (True, False):
    if fast:
      monkeypatch.delenv('CGS_VMC_WIDE_FAST', raising=False)
    else:
      monkeypatch.setenv('CGS_VMC_WIDE_FAST', '0')
    from cgs_vmc_amd.engine import VmcEngine
    rng = np.random.default_rng(0)
    n, h, L, b = 36, 512, 3, 64
    theta = vo.init_params(n, h, L, rng)
    cfg = vo.random_configurations(n, b, np.random.RandomState(1))
    bonds = vo.torus_bonds(6, 6)
    eng = VmcEngine(n, b, L, h, seed=2024)
    assert eng.kernel_path() == (1 if fast else 2)
    eng.set_params(theta); eng.set_configs(cfg); eng.set_bonds(bonds, -1.0, 1.0)
    eng.reset_accumulators()
    eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
    outs.append((eng.amplitude()[0], eng.local_energy()[0], eng.get_accumulators()))
    eng.close()
  (l0, e0, a0), (l1, e1, a1) = outs
  _close(l0, l1, 2e-5)
  _close(e0, e1, 2e-4)
  assert np.abs(a0 - a1).max() < 2e-3 * np.abs(a1).max()


@pytest.mark.parametrize('n,h,L,b,kind', [(16, 1000, 3, 40, 'chain'), (36, 644, 4, 33, 'torus6x6'), (10, 1024, 2, 64, 'chain'),
                                          (16, 640, 3, 200, 'chain'), (16, 800, 2, 130, 'chain')])
def test_general_path_gemm_tilings_agree(monkeypatch, n, h, L, b, kind):
  """The general path's H x H layers on the 128 x 128-tile kernel (k_gemm128, forced with CGS_VMC_GEMM128=2 -- by
  itself only taken for grids of >= 1024 tiles where the ring does not apply) and on the LDS-DMA ring kernel
  (k_gemm_ring, =5: wherever K is a multiple of 32, i.e. the 1024-, 640- and 800-unit cases here -- 32, 20 and 25
  stages: whole and broken turns of the four-slot ring -- with row counts below and off the tile; by itself taken
  from 160 tiles on) against the oracle and against the 64 x 64-tile kernel (CGS_VMC_GEMM128=0): ragged
  row counts, widths that are no multiple of the tile (1000, 644), K tails."""
  from cgs_vmc_amd.engine import VmcEngine
  rng = np.random.default_rng(3)
  theta = vo.init_params(n, h, L, rng)
  theta += (0.02 * rng.standard_normal(theta.size)).astype(np.float32)
  cfg = vo.random_configurations(n, b, np.random.RandomState(4))
  bonds = _bonds(kind, n)
  amp = lambda c: vo.fc_psi(theta, c, h, L, dtype=np.float64)
  e_ref = vo.local_value(amp, cfg, bonds, -1.0, 1.0, dtype=np.float64)
  outs = []
  for mode in ('2', '5', '0'):
    monkeypatch.setenv('CGS_VMC_GEMM128', mode)
    eng = VmcEngine(n, b, L, h, seed=2024)
    assert eng.kernel_path() == 2
    eng.set_params(theta); eng.set_configs(cfg); eng.set_bonds(bonds, -1.0, 1.0)
    logit = eng.amplitude()[0]
    e = eng.local_energy()[0]
    _close(logit, vo.fc_logit(theta, cfg, h, L, dtype=np.float64), 2e-5)
    _close(e, e_ref, 2e-4)
    eng.mc_steps(n)
    outs.append((logit, e, eng.get_configs()))
    eng.close()
  for o in outs[:2]:
    _close(o[0], outs[2][0], 2e-5)
    _close(o[1], outs[2][1], 2e-4)


def test_general_path_full_size_default_dispatch(monkeypatch):
  """bench.py's general-path workload at its full size (10 x 10 torus, 3 x 1024 units, 4096 chains) on the DEFAULT
  dispatch: 4096 rows x 8 column tiles = 256 tiles take k_gemm_ring by themselves (the small shapes above only reach it
  through CGS_VMC_GEMM128=5), the sampler's and the local energies' last layers run the row-dot epilogue, the
  local-energy rows go through 131,072-row blocks.  Logits of every chain and two injected steps against the fp64
  oracle, local energies of sampled chains, then a 100-step sweep twice from the same state and seed: the same
  chains, the same accept count (no float atomics, a fixed order of additions everywhere) -- and the same run with
  the row-dot epilogue and the ring switched off within the logit tolerance."""
  from cgs_vmc_amd.engine import VmcEngine
  n, h, L, b = 100, 1024, 3, 4096
  rng = np.random.default_rng(11)
  theta = vo.init_params(n, h, L, rng)
  theta += (0.01 * rng.standard_normal(theta.size)).astype(np.float32)
  cfg = vo.random_configurations(n, b, np.random.RandomState(12))
  bonds = vo.torus_bonds(10, 10)
  logit_ref = lambda c: vo.fc_logit(theta, c, h, L, dtype=np.float64)
  amp = lambda c: vo.fc_psi(theta, c, h, L, dtype=np.float64)

  def engine():
    e = VmcEngine(n, b, L, h, seed=77)
    assert e.kernel_path() == 2
    e.set_params(theta); e.set_configs(cfg); e.set_bonds(bonds, -1.0, 1.0)
    return e
  eng = engine()
  logit = eng.amplitude()[0]
  _close(logit, logit_ref(cfg), 2e-5)
  e_loc = eng.local_energy()[0]
  idx = np.random.RandomState(5).choice(b, 16, replace=False)
  _close(e_loc[idx], vo.local_value(amp, cfg[idx], bonds, -1.0, 1.0, dtype=np.float64), 2e-4)
  cur = cfg
  for step in range(2):
    u_sites, u_acc = vo.step_uniforms(99, np.arange(b), step, n)
    i_up, i_dn = vo.propose_exchange(cur, u_sites)
    _, acc_ref, ratios = vo.mc_step(amp, cur, i_up, i_dn, u_acc)
    mask = eng.mc_step_injected(i_up, i_dn, u_acc)
    band = np.abs(ratios - np.sqrt(u_acc.astype(np.float64))) < 1e-4 * np.maximum(ratios, 1e-30)
    assert band.sum() < b // 100 and np.array_equal(mask[~band], acc_ref[~band])
    cur = eng.get_configs()
  _close(eng.amplitude()[0], logit_ref(cur), 2e-5)
  eng.set_configs(cfg); eng.step_counter = 0
  acc1 = eng.mc_steps(n)
  got1, e1 = eng.get_configs(), eng.local_energy()[0]
  eng.set_configs(cfg); eng.step_counter = 0
  acc2 = eng.mc_steps(n)
  np.testing.assert_array_equal(eng.get_configs(), got1)
  np.testing.assert_array_equal(eng.local_energy()[0], e1)
  assert acc1 == acc2 and 0.05 * n * b < acc1 < 0.95 * n * b
  assert (got1.sum(axis=1) == 0).all() and (np.abs(got1) == 1).all()
  _close(e1[idx], vo.local_value(amp, got1[idx], bonds, -1.0, 1.0, dtype=np.float64), 2e-4)
  eng.close()
  monkeypatch.setenv('CGS_VMC_ROWDOT', '0'); monkeypatch.setenv('CGS_VMC_GEMM128', '0')
  eng = engine()
  _close(eng.amplitude()[0], logit, 2e-5)
  _close(eng.local_energy()[0], e_loc, 2e-4)
  eng.close()
