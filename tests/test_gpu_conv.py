"""GPU parity of the convolutional ansatz types -- Conv2DNetwork (wavefunctions.py:531-615) and
ResNet2D (wavefunctions.py:710-809) on layers.Conv2dPeriodic / ResBlock2d (layers.py:89-229) --
against the numpy oracle, through the C ABI (csrc/conv.hip: k_conv_rows, k_conv_sweep,
k_conv_back, k_conv_dw).  Tolerances:
  logits: the logit is the fp32 sum of the N * F entries of the last feature map, which cancel
  (|logit| << sum |entries|), so the bound is stated on the summation scale:
  |logit - ref| <= 1e-6 * sum |entries| + 2e-5 (about 16 fp32 ulps of the scale);
  local energies 2e-4 * max(1, |E|), gradient sums 2e-3 * ||.||_inf + 1e-4, accept masks bit-exact
  outside |ratio - sqrt(u)| < 1e-4 ratio, proposals bit-exact (as tests/test_gpu_engine.py).
"""
import numpy as np
import pytest

from oracle import vmc_oracle as vo

pytestmark = pytest.mark.gpu

CONV_SHAPES = [
    # ansatz, size_x, size_y, num_layers / num_blocks, filters, kernel, B, nonlinearity
    ('conv_2d', 4, 4, 2, 4, 3, 20, 'relu'),
    ('conv_2d', 6, 4, 3, 16, 5, 33, 'tanh'),        # non-square: axis 1 = size_x, kernel > size_y
    ('conv_2d', 4, 6, 3, 8, 4, 24, 'sigmoid'),      # even kernel: padding k/2-1 in front, k/2 behind
    ('conv_2d', 5, 5, 2, 5, 2, 17, 'relu'),         # N not a multiple of 4, k = 2: no front padding
    ('conv_2d', 3, 4, 1, 16, 3, 9, 'relu'),         # a single convolution: no hidden activation
    ('conv_2d', 10, 10, 5, 16, 5, 40, 'relu'),      # hparams defaults (utils.py:108-111) on 10 x 10
    ('conv_2d', 4, 4, 3, 16, 1, 16, 'tan'),         # 1 x 1 kernels
    ('conv_2d', 6, 6, 2, 12, 6, 10, 'identity'),    # k = 6 = lattice side
    ('conv_2d', 8, 7, 3, 16, 7, 14, 'relu'),        # k = 7 (49 taps: the largest register-resident kernel)
    ('res_net_2d', 4, 4, 2, 8, 3, 20, 'relu'),
    ('res_net_2d', 6, 6, 2, 16, 5, 40, 'relu'),
    ('res_net_2d', 5, 4, 1, 16, 4, 12, 'relu'),
    ('res_net_2d', 4, 4, 0, 6, 3, 8, 'relu'),       # no block: the initial convolution alone
    ('res_net_2d', 10, 10, 2, 16, 5, 24, 'relu'),   # hparams defaults (utils.py:114) on 10 x 10
    # more than 16 filters: two 16-channel MFMA blocks (conv32.hip)
    ('conv_2d', 4, 4, 3, 32, 3, 20, 'relu'),
    ('conv_2d', 6, 6, 3, 32, 5, 21, 'tanh'),
    ('conv_2d', 10, 10, 4, 24, 5, 26, 'relu'),      # 24 filters padded to 32, 10 x 10
    ('conv_2d', 5, 4, 2, 17, 4, 9, 'sigmoid'),      # 17 filters, even kernel
    ('conv_2d', 6, 6, 2, 20, 6, 7, 'identity'),     # k = 6 with two channel blocks
    ('conv_2d', 7, 8, 2, 24, 7, 9, 'tanh'),         # k = 7 with two channel blocks
    ('res_net_2d', 7, 7, 1, 16, 7, 11, 'relu'),
    ('res_net_2d', 6, 6, 2, 32, 3, 18, 'relu'),
    ('res_net_2d', 4, 6, 1, 28, 5, 13, 'relu'),
    # the cosine (layers.py:15): its derivative needs the pre-activation, which the tape then holds
    ('conv_2d', 4, 4, 3, 8, 3, 20, 'cos'),
    ('conv_2d', 6, 4, 2, 32, 3, 11, 'cos'),
    # round 4: 33 .. 64 filters (three / four channel blocks: conv48.hip, conv64.hip; the weight gradient
    # takes one output block per workgroup) ...
    ('conv_2d', 4, 4, 3, 48, 3, 20, 'relu'),
    ('conv_2d', 6, 6, 3, 64, 5, 13, 'tanh'),
    ('conv_2d', 10, 10, 3, 40, 5, 9, 'relu'),       # 40 filters padded to 48, 10 x 10
    ('conv_2d', 5, 4, 2, 33, 4, 9, 'sigmoid'),      # 33 filters, even kernel
    ('conv_2d', 6, 6, 2, 50, 6, 7, 'identity'),     # 2 x 18 taps per block pair
    ('conv_2d', 7, 8, 2, 64, 7, 6, 'tanh'),         # 25 + 24 taps, the weight gradient in two tap parts
    ('conv_2d', 6, 4, 2, 64, 3, 11, 'cos'),
    ('res_net_2d', 6, 6, 2, 64, 3, 12, 'relu'),
    ('res_net_2d', 4, 6, 1, 44, 5, 10, 'relu'),
    # ... and kernels of 8 and 9 taps per axis (fragments in chunks of <= 25 taps; 16-entry wrap tables at 9)
    ('conv_2d', 8, 8, 2, 8, 8, 11, 'relu'),
    ('conv_2d', 9, 10, 3, 16, 9, 7, 'tanh'),
    ('conv_2d', 6, 4, 2, 12, 9, 9, 'sigmoid'),      # k = 9 on a 6 x 4 lattice: every tap wraps
    ('conv_2d', 8, 8, 2, 24, 8, 6, 'relu'),
    ('conv_2d', 10, 9, 2, 32, 9, 5, 'relu'),
    ('conv_2d', 8, 8, 2, 40, 8, 5, 'tanh'),
    ('conv_2d', 10, 10, 2, 64, 9, 4, 'relu'),       # the largest instantiation
    ('res_net_2d', 8, 8, 1, 16, 9, 8, 'relu'),
    ('res_net_2d', 8, 6, 1, 36, 8, 6, 'relu'),
]
ONE_D = [
    # Conv1DNetwork / ResNet1D (wavefunctions.py:455-527, 618-707): N x 1 lattice, k x 1 taps
    ('conv_1d', 12, 1, 3, 16, 5, 20, 'relu'),
    ('conv_1d', 10, 1, 2, 8, 4, 17, 'tanh'),        # even kernel: k/2 in front, k/2 - 1 behind (layers.py:66-72)
    ('conv_1d', 40, 1, 5, 16, 5, 24, 'relu'),       # hparams defaults (utils.py:98, 108-111)
    ('conv_1d', 9, 1, 2, 5, 2, 11, 'sigmoid'),
    ('res_net_1d', 12, 1, 2, 16, 5, 20, 'relu'),
    ('res_net_1d', 24, 1, 1, 12, 6, 13, 'relu'),
    ('res_net_1d', 40, 1, 2, 16, 3, 16, 'relu'),
    ('conv_1d', 16, 1, 3, 32, 5, 14, 'relu'),
    ('res_net_1d', 12, 1, 1, 24, 3, 10, 'relu'),
    ('conv_1d', 16, 1, 3, 12, 7, 15, 'relu'),       # 7 taps on the chain
    ('res_net_1d', 14, 1, 1, 20, 7, 9, 'relu'),     # ... with two channel blocks
    ('conv_1d', 11, 1, 2, 18, 2, 8, 'tanh'),        # 2 taps, two channel blocks (the variant with the most spills)
    ('conv_1d', 16, 1, 3, 64, 5, 12, 'relu'),
    ('conv_1d', 20, 1, 2, 12, 9, 10, 'relu'),       # 9 taps on the chain
    ('res_net_1d', 18, 1, 1, 48, 8, 7, 'relu'),
]
CONV_SHAPES = CONV_SHAPES + ONE_D
BIG = [
    ('conv_2d', 16, 16, 5, 16, 5, 12, 'relu'),
    ('res_net_2d', 16, 16, 2, 16, 5, 10, 'relu'),
]
IDS = ['{}-{}x{}-L{}-F{}-K{}-B{}-{}'.format(*s) for s in CONV_SHAPES + BIG]


def _make(ansatz, sx, sy, L, f, k, b, nonlin, seed=0, output_activation='exp', noise=0.03):
  from cgs_vmc_amd.engine import VmcEngine
  n = sx * sy
  geom = (f, k, sx, sy)
  rng = np.random.default_rng(seed)
  theta = vo.conv_init_params(ansatz, geom, L, rng)
  if f > 32 or k > 7:
    # the round-4 shapes: up to 81 x 64 inputs per output -- the per-weight noise is scaled to the
    # fan-in so that the logits stay at the magnitudes of the smaller shapes
    noise = min(noise, 0.03 * np.sqrt(400.0 / (f * k * (1 if ansatz in vo.CONV_1D else k))))
  theta += (noise * rng.standard_normal(theta.size)).astype(np.float32)   # non-zero biases
  cfg = vo.random_configurations(n, b, np.random.RandomState(seed + 1))
  # site = a2 + size_y * a1 (row-major reshape); the 1-D types live on the periodic chain
  bonds = vo.chain_bonds(n) if ansatz in vo.CONV_1D else vo.torus_bonds(sy, sx)
  eng = VmcEngine(n, b, L, f, nonlinearity=nonlin, output_activation=output_activation, seed=2024,
                  ansatz=ansatz, kernel_size=k, size_x=sx, size_y=sy)
  assert eng.num_params == theta.size == vo.conv_num_params(ansatz, geom, L)
  eng.set_params(theta)
  eng.set_configs(cfg)
  eng.set_bonds(bonds, -1.0, 1.0)
  return eng, theta, cfg, bonds, geom


def _close(a, b, rel, floor=1.0):
  a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
  tol = rel * np.maximum(floor, np.abs(b))
  bad = np.abs(a - b) > tol
  assert not bad.any(), 'max err {} at {} (tol {})'.format(
      np.abs(a - b).max(), np.argmax(np.abs(a - b)), tol[np.argmax(np.abs(a - b))])


def _logits_close(got, theta, cfg, ansatz, geom, L, nonlin):
  ref, scale = vo.conv_forward(theta, cfg, ansatz, geom, L, nonlin, np.float64, return_tape='scale')
  err = np.abs(np.asarray(got, np.float64) - ref)
  tol = 1e-6 * scale + 2e-5
  assert (err <= tol).all(), 'max err {} at {} (tol {})'.format(err.max(), err.argmax(), tol[err.argmax()])


@pytest.mark.parametrize('ansatz,sx,sy,L,f,k,b,nonlin', CONV_SHAPES + BIG, ids=IDS)
def test_conv_amplitude_and_local_energy(ansatz, sx, sy, L, f, k, b, nonlin):
  eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, nonlin)
  logit_fn, psi_fn = vo.ANSATZ[ansatz][1], vo.ANSATZ[ansatz][0]
  logit, psi = eng.amplitude(cfg)
  _logits_close(logit, theta, cfg, ansatz, geom, L, nonlin)
  _logits_close(eng.amplitude()[0], theta, cfg, ansatz, geom, L, nonlin)             # cached path
  with np.errstate(over='ignore'):      # (psi = inf where the float32 exponential overflows: compared as such)
    np.testing.assert_allclose(psi, np.exp(logit.astype(np.float32) + np.float32(10.0)), rtol=1e-6)
  c2 = vo.random_configurations(sx * sy, 45, np.random.RandomState(9))   # ragged last row group
  _logits_close(eng.amplitude(c2)[0], theta, c2, ansatz, geom, L, nonlin)
  amp = lambda c: psi_fn(theta, c, geom, L, nonlinearity=nonlin, dtype=np.float64)
  for jx in (-1.0, 1.0):
    eng.set_bonds(bonds, jx, 1.0)
    _close(eng.local_energy()[0], vo.local_value(amp, cfg, bonds, jx, 1.0, dtype=np.float64), 2e-4)
  eng.close()


@pytest.mark.parametrize('ansatz,sx,sy,L,f,k,b,nonlin', CONV_SHAPES + BIG[:1], ids=IDS[:len(CONV_SHAPES) + 1])
def test_conv_proposals_injected_step_and_trajectory(ansatz, sx, sy, L, f, k, b, nonlin):
  eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, nonlin)
  n = sx * sy
  logit_fn, psi_fn = vo.ANSATZ[ansatz][1], vo.ANSATZ[ansatz][0]
  amp = lambda c: psi_fn(theta, c, geom, L, nonlinearity=nonlin, dtype=np.float64)
  # proposals of the sampler's own Philox stream: bit-exact (graph_builders.py:59-65)
  for step in (0, 7):
    u_sites, u_acc = vo.step_uniforms(2024, np.arange(b), step, n)
    i_up, i_dn = vo.propose_exchange(cfg, u_sites)
    g_up, g_dn, g_u = eng.debug_proposals(step)
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
    _logits_close(eng.amplitude()[0], theta, cur, ansatz, geom, L, nonlin)
  # 8 steps of the persistent sampler against the oracle trajectory
  eng.step_counter = 0
  start = cur.copy()
  ok = np.ones(b, bool)
  for step in range(8):
    u_sites, u_acc = vo.step_uniforms(2024, np.arange(b), step, n)
    i_up, i_dn = vo.propose_exchange(cur, u_sites)
    cur, acc, ratios = vo.mc_step(amp, cur, i_up, i_dn, u_acc)
    ok &= ~(np.abs(ratios - np.sqrt(u_acc.astype(np.float64))) < 1e-4 * np.maximum(ratios, 1e-30))
  accepted = eng.mc_steps(8)
  got = eng.get_configs()
  np.testing.assert_array_equal(got[ok], cur[ok])
  assert ok.sum() > b // 2 and (got.sum(1) == start.sum(1)).all() and 0 <= accepted <= 8 * b
  _logits_close(eng.amplitude()[0], theta, got, ansatz, geom, L, nonlin)
  _close(eng.local_energy()[0], vo.local_value(amp, got, bonds, -1.0, 1.0, dtype=np.float64), 2e-4)
  eng.close()


@pytest.mark.parametrize('ansatz,sx,sy,L,f,k,b,nonlin', CONV_SHAPES + BIG, ids=IDS)
def test_conv_energy_gradient_accumulators(ansatz, sx, sy, L, f, k, b, nonlin):
  from cgs_vmc_amd import _hip
  eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, nonlin)
  acc = vo.Accumulators(theta.size, np.float64)
  eng.reset_accumulators()
  vo.energy_gradient_accumulate(acc, theta, cfg, bonds, -1.0, 1.0, -10.0, geom, L, np.float64,
                                ansatz=ansatz, nonlinearity=nonlin)
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  eng.mc_steps(3)                                       # second batch on moved chains
  cur = eng.get_configs()
  vo.energy_gradient_accumulate(acc, theta, cur, bonds, -1.0, 1.0, -10.0, geom, L, np.float64,
                                ansatz=ansatz, nonlinearity=nonlin)
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  got = eng.get_accumulators()
  p = theta.size
  for name, g, r in (('g1', got[:p], acc.g1_total), ('g2', got[p:2 * p], acc.g2_total)):
    tol = 2e-3 * np.abs(r).max() + 1e-4
    assert np.abs(g - r).max() < tol, (name, np.abs(g - r).max(), tol, int(np.argmax(np.abs(g - r))))
  sc = got[2 * p:]
  assert abs(sc[0] - acc.e_total) < 2e-4 * max(1, abs(acc.e_total)) and sc[1] == 2 * b and sc[4] == 2
  grad_ref = vo.energy_gradient(acc)
  grad = eng.get_gradient(_hip.VMC_MODE_ENERGY_GRADIENT)
  # grad = mean(g2) - mean(E) mean(g1) cancels (exactly, for the 1 x 1 kernels and for the linear
  # network whose kernel spans the lattice: psi is then constant at fixed magnetisation), so the
  # fp32 bound carries the scale of the two terms
  cancel = 4e-6 * max(np.abs(acc.g2_total).max(), abs(acc.mean_energy()) * np.abs(acc.g1_total).max()) / 2
  assert np.abs(grad - grad_ref).max() < 2e-3 * np.abs(grad_ref).max() + 2e-4 + cancel
  st = vo.AdamState(p)
  th_ref = vo.adam_apply(st, theta, grad, 1e-3, 0.9, 0.99, 1e-8)
  eng.apply_adam(_hip.VMC_MODE_ENERGY_GRADIENT, 1e-3, 0.9, 0.99, 1e-8)
  np.testing.assert_allclose(eng.get_params(), th_ref, rtol=0, atol=2e-6)   # Adam of the GPU's own gradient
  _logits_close(eng.amplitude()[0], eng.get_params(), cur, ansatz, geom, L, nonlin)
  eng.close()


@pytest.mark.parametrize('ansatz,sx,sy,L,f,k,b,nonlin', [
    ('conv_2d', 6, 5, 3, 16, 3, 37, 'relu'),
    ('conv_2d', 5, 6, 2, 24, 4, 21, 'cos'),           # two channel blocks, even kernel, pre-activation tape
    ('res_net_2d', 4, 6, 2, 8, 5, 19, 'relu'),
    ('conv_1d', 13, 1, 3, 12, 5, 14, 'tanh'),
])
def test_conv_weight_gradient_in_row_bands(ansatz, sx, sy, L, f, k, b, nonlin, monkeypatch):
  """Lattices whose padded sample does not fit the LDS of the weight-gradient kernel are staged in
  bands of rows (conv.hip: launch_conv_dw).  CGS_VMC_CONV_DW_BAND forces bands on small lattices: bands
  of 1, 2 and 4 rows (a short last band included) give the sums of the unbanded run up to the order of
  summation, and both match the oracle."""
  from cgs_vmc_amd import _hip
  outs = []
  for band in (None, 1, 2, 4):
    if band is None:
      monkeypatch.delenv('CGS_VMC_CONV_DW_BAND', raising=False)
    else:
      monkeypatch.setenv('CGS_VMC_CONV_DW_BAND', str(band))
    eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, nonlin)
    eng.reset_accumulators()
    eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
    outs.append(eng.get_accumulators())
    eng.close()
  acc = vo.Accumulators(theta.size, np.float64)
  vo.energy_gradient_accumulate(acc, theta, cfg, bonds, -1.0, 1.0, -10.0, geom, L, np.float64,
                                ansatz=ansatz, nonlinearity=nonlin)
  p = theta.size
  for got in outs:
    for name, g, r in (('g1', got[:p], acc.g1_total), ('g2', got[p:2 * p], acc.g2_total)):
      tol = 2e-3 * np.abs(r).max() + 1e-4
      assert np.abs(g - r).max() < tol, (name, np.abs(g - r).max(), tol)
  for got in outs[1:]:
    assert np.abs(got[:2 * p] - outs[0][:2 * p]).max() <= 2e-5 * np.abs(outs[0][:2 * p]).max()


@pytest.mark.parametrize('sx,sy,f,k', [(32, 32, 16, 5), (24, 24, 32, 5), (30, 36, 16, 3)])
def test_conv_weight_gradient_on_the_largest_lattices(sx, sy, f, k):
  """The largest lattices the forward kernels take (one sample's feature maps in LDS): their padded
  samples exceed the weight-gradient kernel's LDS, which then walks them in bands of rows on its own.
  A handful of bonds keeps the oracle's local energies cheap; the sums are over every site."""
  from cgs_vmc_amd import _hip
  L, b, ansatz, nonlin = 2, 5, 'conv_2d', 'relu'
  eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, nonlin, noise=0.01)
  bonds = [bonds[i] for i in range(0, len(bonds), max(1, len(bonds) // 12))][:12]
  eng.set_bonds(bonds, -1.0, 1.0)
  theta = (0.5 * theta).astype(np.float32)            # keeps exp(logit) of ~1000 sites inside fp64 for the oracle
  eng.set_params(theta)
  acc = vo.Accumulators(theta.size, np.float64)
  eng.reset_accumulators()
  vo.energy_gradient_accumulate(acc, theta, cfg, bonds, -1.0, 1.0, -10.0, geom, L, np.float64,
                                ansatz=ansatz, nonlinearity=nonlin)
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  got = eng.get_accumulators()
  p = theta.size
  for name, g, r in (('g1', got[:p], acc.g1_total), ('g2', got[p:2 * p], acc.g2_total)):
    tol = 2e-3 * np.abs(r).max() + 1e-4
    assert np.abs(g - r).max() < tol, (name, np.abs(g - r).max(), tol)
  eng.close()


@pytest.mark.parametrize('ansatz', ['conv_2d', 'res_net_2d'])
def test_conv_log_overlap_itswo_accumulators(ansatz):
  from cgs_vmc_amd import _hip
  sx, sy, L, f, k, b = 4, 4, 2, 8, 3, 48
  eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, 'relu')
  eng.transfer_params()
  rng = np.random.default_rng(8)
  theta2 = theta + (0.02 * rng.standard_normal(theta.size)).astype(np.float32)
  eng.set_params(theta2)
  eng.set_shift(-9.0)
  acc = vo.Accumulators(theta.size, np.float64)
  vo.log_overlap_accumulate(acc, theta2, theta, cfg, bonds, -1.0, 1.0, -9.0, -10.0, 0.12, geom, L,
                            np.float64, ansatz=ansatz)
  eng.reset_accumulators()
  eng.accumulate(_hip.VMC_MODE_LOG_OVERLAP_ITSWO, 0.12)
  grad_ref = vo.log_overlap_gradient(acc)
  grad = eng.get_gradient(_hip.VMC_MODE_LOG_OVERLAP_ITSWO)
  assert np.abs(grad - grad_ref).max() < 2e-3 * np.abs(grad_ref).max() + 2e-4
  eng.close()


@pytest.mark.parametrize('oact', ['tanh', 'identity', 'sigmoid'])
def test_conv_non_exp_output_activation(oact):
  """wavefunctions.py:576-579: any output activation but exp gives psi = g(sum), no shift."""
  from cgs_vmc_amd import _hip
  ansatz, sx, sy, L, f, k, b = 'conv_2d', 4, 4, 2, 8, 3, 32
  eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, 'tanh', output_activation=oact,
                                       noise=0.01)
  amp = lambda c: vo.ANSATZ[ansatz][0](theta, c, geom, L, nonlinearity='tanh',
                                       output_activation=oact, dtype=np.float64)
  logit, psi = eng.amplitude(cfg)
  _logits_close(logit, theta, cfg, ansatz, geom, L, 'tanh')
  np.testing.assert_allclose(psi, vo.NONLINEARITIES[oact](logit.astype(np.float64)), rtol=1e-5, atol=1e-6)
  _close(eng.local_energy()[0], vo.local_value(amp, cfg, bonds, -1.0, 1.0, dtype=np.float64), 5e-4)
  acc = vo.Accumulators(theta.size, np.float64)
  vo.energy_gradient_accumulate(acc, theta, cfg, bonds, -1.0, 1.0, -10.0, geom, L, np.float64,
                                ansatz=ansatz, nonlinearity='tanh', output_activation=oact)
  eng.reset_accumulators()
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  grad_ref = vo.energy_gradient(acc)
  grad = eng.get_gradient(_hip.VMC_MODE_ENERGY_GRADIENT)
  assert np.abs(grad - grad_ref).max() < 3e-3 * np.abs(grad_ref).max() + 2e-4
  eng.close()


def test_conv_shard_invariance_and_reproducibility():
  """Chains [16, 48) of a 64-chain run walk the same trajectory as a 32-chain shard with
  chain_offset 16 (Philox keyed by the global chain id); two identical runs agree bit for bit."""
  from cgs_vmc_amd.engine import VmcEngine
  ansatz, sx, sy, L, f, k = 'conv_2d', 6, 6, 3, 16, 3
  n, geom = sx * sy, (16, 3, 6, 6)
  theta = vo.conv_init_params(ansatz, geom, L, np.random.default_rng(3))
  cfg = vo.random_configurations(n, 64, np.random.RandomState(4))
  outs = []
  for (b, off, rows) in ((64, 0, slice(0, 64)), (64, 0, slice(0, 64)), (32, 16, slice(16, 48))):
    eng = VmcEngine(n, b, L, f, seed=11, ansatz=ansatz, kernel_size=k, size_x=sx, size_y=sy,
                    chain_offset=off)
    eng.set_params(theta); eng.set_configs(cfg[rows]); eng.set_bonds(vo.torus_bonds(6, 6), -1.0, 1.0)
    eng.mc_steps(n)
    outs.append((eng.get_configs(), eng.local_energy()[0]))
    eng.close()
  np.testing.assert_array_equal(outs[0][0], outs[1][0])
  np.testing.assert_array_equal(outs[0][1], outs[1][1])
  np.testing.assert_array_equal(outs[0][0][16:48], outs[2][0])
  np.testing.assert_array_equal(outs[0][1][16:48], outs[2][1])


def test_conv_error_behaviour():
  from cgs_vmc_amd.engine import VmcEngine
  kw = dict(ansatz='conv_2d', kernel_size=3, size_x=4, size_y=4)
  with pytest.raises(ValueError):
    VmcEngine(15, 8, 2, 8, **kw)                          # size_x * size_y != num_sites
  eng = VmcEngine(16, 8, 2, 65, **kw)                     # more than 64 filters: the general path (conv_general.hip)
  assert eng.kernel_path() == 6
  eng.close()
  with pytest.raises(NotImplementedError):
    VmcEngine(16, 8, 2, 1025, **kw)
  with pytest.raises(NotImplementedError):
    VmcEngine(16, 8, 2, 8, ansatz='conv_2d', kernel_size=10, size_x=4, size_y=4)    # lattice side < kernel_size / 2
  with pytest.raises(NotImplementedError):
    VmcEngine(16 * 16, 8, 2, 8, ansatz='conv_2d', kernel_size=32, size_x=16, size_y=16)
  eng = VmcEngine(16, 8, 2, 8, output_activation='tanh', **kw)
  with pytest.raises(NotImplementedError):
    eng.sr_reserve(2)                                     # SR (an extension) needs the exp output
  eng.close()


def test_conv_through_run_training_and_evaluation(tmp_path):
  """--wavefunction_type=conv_2d / res_net_2d through the run_training / run_energy_evaluation
  counterparts: Sonnet variable names, energy of the 4x4 torus well below the Neel value."""
  import os
  from cgs_vmc_amd import lattice, run_energy_evaluation, run_training, session as session_lib
  from cgs_vmc_amd import wavefunctions
  for wf_type, var in (('conv_2d', 'conv_2d_network/conv_2d_periodic_1/conv_2d/w'),
                       ('res_net_2d', 'res_net_2d/res_block_2d/second_conv/conv_2d/b')):
    session_lib.reset_default_graph()
    wavefunctions.reset_name_scope()
    os.environ.update(CGS_VMC_SEED='77', CGS_VMC_CONFIG_SEED='5', CGS_VMC_INIT_SEED='31')
    d = str(tmp_path / wf_type)
    os.makedirs(d)
    lattice.write_bonds(d, lattice.torus_bonds(4, 4))
    hp = ('batch_size=256,size_x=4,size_y=4,num_conv_layers=2,num_resnet_blocks=1,num_conv_filters=8,'
          'kernel_size=3,num_equilibration_sweeps=10,num_batches_per_epoch=8,'
          'learning_rates=[0.003,0.001],learning_rate_stops=[60]')
    run_training.main(['--checkpoint_dir', d, '--num_sites', '16', '--heisenberg_jx', '-1.0',
                       '--wavefunction_type', wf_type, '--optimizer', 'EnergyGradient',
                       '--num_epochs', '80', '--hparams', hp])
    energies = [float(x) for x in open(os.path.join(d, 'metrics.txt')).read().split()]
    tail = np.mean(energies[-10:])
    assert -11.2285 - 0.05 < tail < -9.0, (wf_type, tail, energies[::10])
    ck = session_lib.latest_checkpoint(d)
    assert var in set(np.load(ck + '.npz').files)
    session_lib.reset_default_graph()
    wavefunctions.reset_name_scope()
    run_energy_evaluation.main(['--checkpoint_dir', d, '--heisenberg_jx', '-1.0',
                                '--hparams', 'num_evaluation_samples=5'])


def test_conv_1d_through_run_training(tmp_path):
  """--wavefunction_type=conv_1d on the reference's default lattice, the periodic chain
  (run_training.py:103-109: no J.txt): 16 sites, E0 = -7.1423 (exact diagonalisation)."""
  import os
  from cgs_vmc_amd import run_training, session as session_lib, wavefunctions
  session_lib.reset_default_graph()
  wavefunctions.reset_name_scope()
  os.environ.update(CGS_VMC_SEED='77', CGS_VMC_CONFIG_SEED='5', CGS_VMC_INIT_SEED='31')
  d = str(tmp_path)
  hp = ('batch_size=256,num_conv_layers=3,num_conv_filters=8,kernel_size=4,num_equilibration_sweeps=10,'
        'num_batches_per_epoch=8,learning_rates=[0.003,0.001],learning_rate_stops=[60]')
  run_training.main(['--checkpoint_dir', d, '--num_sites', '16', '--heisenberg_jx', '-1.0',
                     '--wavefunction_type', 'conv_1d', '--optimizer', 'EnergyGradient',
                     '--num_epochs', '80', '--hparams', hp])
  energies = [float(x) for x in open(os.path.join(d, 'metrics.txt')).read().split()]
  tail = np.mean(energies[-10:])
  assert -7.1423 - 0.05 < tail < -6.0, (tail, energies[::10])
  ck = session_lib.latest_checkpoint(d)
  assert 'conv_1d_network/conv_1d_periodic_2/conv_1d/w' in set(np.load(ck + '.npz').files)


def _random_conv_shapes(count, seed=2025, kmax=6, with_cos=False, fmax=32):
  """Seeded random geometries inside the limits vmc_create states (kernel 1..9, <= 64 filters,
  lattice sides >= kernel // 2): every padding parity, ragged batches, k larger than a side."""
  rng = np.random.default_rng(seed)
  # (tan and exp hidden units are covered by CONV_SHAPES at controlled magnitudes: near a pole of
  # tan the fp32 error of the pre-activation is amplified without bound)
  acts = ['relu', 'tanh', 'sigmoid', 'identity'] + (['cos'] if with_cos else [])
  shapes = []
  while len(shapes) < count:
    ansatz = ['conv_2d', 'res_net_2d', 'conv_1d', 'res_net_1d'][int(rng.integers(4))]
    k = int(rng.integers(1, kmax + 1))
    if ansatz in vo.CONV_1D:
      sx, sy = int(rng.integers(max(2, k // 2), 33)), 1
    else:
      sx, sy = int(rng.integers(max(2, k // 2), 9)), int(rng.integers(max(2, k // 2), 9))
    if (sx * sy) % 2 or sx * sy < 4:
      continue
    resnet = ansatz.startswith('res_net')
    L = int(rng.integers(0, 3)) if resnet else int(rng.integers(1, 5))
    f = int(rng.integers(1, 17)) if len(shapes) % 3 else int(rng.integers(17, fmax + 1))
    b = int(rng.integers(1, 41))
    nonlin = 'relu' if resnet else acts[int(rng.integers(len(acts)))]
    shapes.append((ansatz, sx, sy, L, f, k, b, nonlin))
  return shapes


# the round-2 / early round-3 set (kernel <= 6), then one with 7 x 7 kernels and cos, then (round 4) one up to
# 9 x 9 kernels and 64 filters
RANDOM_SHAPES = (_random_conv_shapes(36) + _random_conv_shapes(18, seed=3031, kmax=7, with_cos=True) +
                 _random_conv_shapes(24, seed=4044, kmax=9, with_cos=True, fmax=64))


@pytest.mark.parametrize('ansatz,sx,sy,L,f,k,b,nonlin', RANDOM_SHAPES,
                         ids=['{}-{}x{}-L{}-F{}-K{}-B{}-{}'.format(*s) for s in RANDOM_SHAPES])
def test_conv_random_shapes(ansatz, sx, sy, L, f, k, b, nonlin):
  """Amplitudes, local energies, one injected mc_step and the gradient sums on random geometries."""
  from cgs_vmc_amd import _hip
  # the parameter noise is per weight: with up to 36 x 32 inputs per output it is scaled down so that
  # the residual stacks stay at logits of order 10, not 10^27
  eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, nonlin, seed=b + 7 * k,
                                       noise=0.03 if f <= 16 else (0.01 if f <= 32 else 0.005))
  n = sx * sy
  psi_fn = vo.ANSATZ[ansatz][0]
  amp = lambda c: psi_fn(theta, c, geom, L, nonlinearity=nonlin, dtype=np.float64)
  _logits_close(eng.amplitude()[0], theta, cfg, ansatz, geom, L, nonlin)
  e_ref = vo.local_value(amp, cfg, bonds, -1.0, 1.0, dtype=np.float64)
  # an amplitude ratio carries the fp32 rounding of two logits, each ~ 1e-6 of the sum of the magnitudes of
  # its terms (what _logits_close allows): with 64 filters that sum reaches several thousand
  _, scale = vo.conv_forward(theta, cfg, ansatz, geom, L, nonlin, np.float64, return_tape='scale')
  _close(eng.local_energy()[0], e_ref, max(2e-4, 4e-6 * float(np.max(scale))))
  acc = vo.Accumulators(theta.size, np.float64)
  vo.energy_gradient_accumulate(acc, theta, cfg, bonds, -1.0, 1.0, -10.0, geom, L, np.float64,
                                ansatz=ansatz, nonlinearity=nonlin)
  eng.reset_accumulators()
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  got = eng.get_accumulators()
  p = theta.size
  for name, g, r in (('g1', got[:p], acc.g1_total), ('g2', got[p:2 * p], acc.g2_total)):
    tol = 2e-3 * np.abs(r).max() + 1e-4
    assert np.abs(g - r).max() < tol, (name, np.abs(g - r).max(), tol, int(np.argmax(np.abs(g - r))))
  u_sites, u_acc = vo.step_uniforms(5, np.arange(b), 0, n)
  i_up, i_dn = vo.propose_exchange(cfg, u_sites)
  _, acc_ref, ratios = vo.mc_step(amp, cfg, i_up, i_dn, u_acc)
  mask = eng.mc_step_injected(i_up, i_dn, u_acc)
  band = np.abs(ratios - np.sqrt(u_acc.astype(np.float64))) < 1e-4 * np.maximum(ratios, 1e-30)
  assert np.array_equal(mask[~band], acc_ref[~band])
  eng.close()


@pytest.mark.parametrize('wf_type,e0', [('conv_2d', -11.2285), ('res_net_1d', -7.1423)])
def test_conv_log_overlap_itswo_training_entry(tmp_path, wf_type, e0):
  """--optimizer=LogOverlapITSWO with a convolutional ansatz (vmc_epoch_log_overlap on the conv
  kernels): 40 epochs approach the exact ground-state energy of the 4x4 torus / the 16-site ring
  from above; --optimizer=StochasticReconfiguration (an extension; round 3: also over the
  convolutional kernels) then lowers the energy further from the same checkpoint directory's start."""
  import os
  from cgs_vmc_amd import lattice, run_training, session as session_lib, wavefunctions
  session_lib.reset_default_graph()
  wavefunctions.reset_name_scope()
  os.environ.update(CGS_VMC_SEED='77', CGS_VMC_CONFIG_SEED='5', CGS_VMC_INIT_SEED='31')
  d = str(tmp_path)
  lattice.write_bonds(d, lattice.chain_bonds(16) if wf_type.endswith('1d') else lattice.torus_bonds(4, 4))
  hp = ('batch_size=256,size_x=4,size_y=4,num_conv_layers=2,num_resnet_blocks=1,num_conv_filters=8,'
        'kernel_size=3,num_equilibration_sweeps=10,num_batches_per_epoch=8,'
        'learning_rates=[0.003,0.001],learning_rate_stops=[60]')
  args = ['--checkpoint_dir', d, '--num_sites', '16', '--heisenberg_jx', '-1.0',
          '--wavefunction_type', wf_type, '--num_epochs', '40', '--hparams', hp]
  run_training.main(args + ['--optimizer', 'LogOverlapITSWO'])
  energies = [float(x) for x in open(os.path.join(d, 'metrics.txt')).read().split()]
  tail = np.mean(energies[-5:])
  assert e0 - 0.05 < tail < 0.95 * e0, (wf_type, tail, energies[::8])
  session_lib.reset_default_graph()
  wavefunctions.reset_name_scope()
  d2 = os.path.join(d, 'sr')
  os.makedirs(d2)
  lattice.write_bonds(d2, lattice.chain_bonds(16) if wf_type.endswith('1d') else lattice.torus_bonds(4, 4))
  hp_sr = hp.replace('learning_rates=[0.003,0.001]', 'learning_rates=[0.02,0.01]')
  run_training.main(['--checkpoint_dir', d2, '--num_sites', '16', '--heisenberg_jx', '-1.0', '--wavefunction_type',
                     wf_type, '--num_epochs', '30', '--hparams', hp_sr, '--optimizer', 'StochasticReconfiguration'])
  e_sr = [float(x) for x in open(os.path.join(d2, 'metrics.txt')).read().split()]
  assert np.isfinite(e_sr).all() and np.mean(e_sr[-5:]) < e_sr[0] - 0.5 and np.mean(e_sr[-5:]) > e0 - 0.05, e_sr[::5]
