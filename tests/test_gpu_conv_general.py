"""GPU parity of the GENERAL convolution path (csrc/conv_general.hip, kernel_path() == 6) against the numpy oracle,
through the C ABI: Conv2DNetwork / ResNet2D / Conv1DNetwork / ResNet1D (wavefunctions.py:455-809 on
layers.Conv*Periodic / ResBlock*, layers.py:24-293) at shapes the fused kernels refuse -- num_conv_filters > 64,
kernel_size > 9, feature maps beyond 160 KiB of LDS; the reference takes any value (utils.py:107-111) -- and, forced
with CGS_VMC_CONV_GENERAL=1, at shapes both paths take, where the two must agree.  Amplitudes, local energies,
proposals, injected steps, trajectories, the gradient accumulators and (single-rank) stochastic reconfiguration.
Tolerances as tests/test_gpu_conv.py."""
import numpy as np
import pytest

from cgs_vmc_amd import _hip
from oracle import vmc_oracle as vo
from tests.test_gpu_conv import _close, _logits_close, _make

pytestmark = pytest.mark.gpu

GENERAL_SHAPES = [
    # ansatz, size_x, size_y, num_layers / num_blocks, filters, kernel, B, nonlinearity
    ('conv_2d', 4, 4, 2, 80, 3, 12, 'relu'),        # more than 64 filters
    ('conv_2d', 6, 6, 3, 72, 3, 9, 'tanh'),
    ('conv_2d', 4, 4, 2, 130, 3, 6, 'relu'),        # filters no multiple of 4: the scalar gather, the strided GEMM
    ('conv_2d', 6, 6, 2, 128, 3, 600, 'relu'),      # 21,600 x 1152 x 128: the second convolution runs on k_gemm_ring
    ('conv_2d', 10, 10, 2, 8, 11, 7, 'relu'),       # kernel_size > 9
    ('conv_2d', 12, 12, 2, 6, 10, 5, 'sigmoid'),    # ... even: k/2 - 1 in front, k/2 behind
    ('conv_2d', 20, 20, 2, 64, 3, 4, 'relu'),       # two maps of 102 KB: beyond the LDS of the fused kernels
    ('conv_2d', 36, 36, 3, 16, 5, 2, 'cos'),       # 1,296 sites at 16 filters: two maps of 83 KB (beyond the 160 KiB with the rest)
    ('res_net_2d', 4, 4, 2, 96, 3, 8, 'relu'),
    ('res_net_2d', 8, 8, 1, 12, 10, 6, 'relu'),
    ('res_net_2d', 4, 6, 0, 70, 3, 5, 'relu'),      # no block: the initial convolution alone
    ('conv_1d', 30, 1, 3, 100, 5, 9, 'relu'),
    ('conv_1d', 26, 1, 2, 10, 12, 8, 'tanh'),       # even 1-D kernel: k/2 in front, k/2 - 1 behind
    ('res_net_1d', 24, 1, 2, 70, 3, 7, 'relu'),
]
IDS = ['{}-{}x{}-L{}-F{}-K{}-B{}-{}'.format(*s) for s in GENERAL_SHAPES]
# shapes both paths take (a subset of tests/test_gpu_conv.py's)
BOTH = [
    ('conv_2d', 6, 4, 3, 16, 5, 33, 'tanh'),
    ('conv_2d', 4, 6, 3, 8, 4, 24, 'sigmoid'),
    ('conv_2d', 5, 5, 2, 5, 2, 17, 'relu'),
    ('conv_2d', 3, 4, 1, 16, 3, 9, 'relu'),
    ('conv_2d', 10, 10, 5, 16, 5, 40, 'relu'),
    ('conv_2d', 6, 6, 3, 64, 5, 13, 'tanh'),
    ('conv_2d', 6, 4, 2, 12, 9, 9, 'sigmoid'),
    ('res_net_2d', 6, 6, 2, 16, 5, 40, 'relu'),
    ('res_net_2d', 5, 4, 1, 16, 4, 12, 'relu'),
    ('conv_1d', 10, 1, 2, 8, 4, 17, 'tanh'),
    ('res_net_1d', 24, 1, 1, 12, 6, 13, 'relu'),
]


def _check_forward_and_sampler(eng, theta, cfg, bonds, geom, ansatz, L, nonlin, b, steps=6):
  sx, sy = geom[2], geom[3]
  n = sx * sy
  psi_fn = vo.ANSATZ[ansatz][0]
  logit, psi = eng.amplitude(cfg)
  _logits_close(logit, theta, cfg, ansatz, geom, L, nonlin)
  _logits_close(eng.amplitude()[0], theta, cfg, ansatz, geom, L, nonlin)             # cached path
  with np.errstate(over='ignore'):
    np.testing.assert_allclose(psi, np.exp(logit.astype(np.float32) + np.float32(10.0)), rtol=1e-6)
  c2 = vo.random_configurations(n, 21, np.random.RandomState(9))
  _logits_close(eng.amplitude(c2)[0], theta, c2, ansatz, geom, L, nonlin)
  amp = lambda c: psi_fn(theta, c, geom, L, nonlinearity=nonlin, dtype=np.float64)
  for jx in (-1.0, 1.0):
    eng.set_bonds(bonds, jx, 1.0)
    _close(eng.local_energy()[0], vo.local_value(amp, cfg, bonds, jx, 1.0, dtype=np.float64), 2e-4)
  u_sites, u_acc = vo.step_uniforms(2024, np.arange(b), 7, n)
  i_up, i_dn = vo.propose_exchange(cfg, u_sites)
  g_up, g_dn, g_u = eng.debug_proposals(7)
  np.testing.assert_array_equal(g_up, i_up); np.testing.assert_array_equal(g_dn, i_dn)
  np.testing.assert_array_equal(g_u, u_acc)
  cur = cfg
  for step in range(2):
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
  eng.step_counter = 0
  start = cur.copy()
  ok = np.ones(b, bool)
  for step in range(steps):
    u_sites, u_acc = vo.step_uniforms(2024, np.arange(b), step, n)
    i_up, i_dn = vo.propose_exchange(cur, u_sites)
    cur, acc, ratios = vo.mc_step(amp, cur, i_up, i_dn, u_acc)
    ok &= ~(np.abs(ratios - np.sqrt(u_acc.astype(np.float64))) < 1e-4 * np.maximum(ratios, 1e-30))
  accepted = eng.mc_steps(steps)
  got = eng.get_configs()
  np.testing.assert_array_equal(got[ok], cur[ok])
  assert ok.sum() > b // 2 and (got.sum(1) == start.sum(1)).all() and 0 <= accepted <= steps * b
  _close(eng.local_energy()[0], vo.local_value(amp, got, bonds, 1.0, 1.0, dtype=np.float64), 2e-4)


@pytest.mark.parametrize('ansatz,sx,sy,L,f,k,b,nonlin', GENERAL_SHAPES, ids=IDS)
def test_general_convolution_beyond_the_fused_limits(ansatz, sx, sy, L, f, k, b, nonlin):
  eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, nonlin)
  assert eng.kernel_path() == 6
  _check_forward_and_sampler(eng, theta, cfg, bonds, geom, ansatz, L, nonlin, b, steps=4 if b > 100 else 6)
  eng.close()


@pytest.mark.parametrize('ansatz,sx,sy,L,f,k,b,nonlin', BOTH)
def test_general_and_fused_convolution_paths_agree(monkeypatch, ansatz, sx, sy, L, f, k, b, nonlin):
  monkeypatch.delenv('CGS_VMC_CONV_GENERAL', raising=False)
  eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, nonlin)
  assert eng.kernel_path() == 3
  fused = (eng.amplitude()[0], eng.local_energy()[0])
  eng.close()
  monkeypatch.setenv('CGS_VMC_CONV_GENERAL', '1')
  eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, nonlin)
  assert eng.kernel_path() == 6
  _, scale = vo.conv_forward(theta, cfg, ansatz, geom, L, nonlin, np.float64, return_tape='scale')
  assert (np.abs(eng.amplitude()[0].astype(np.float64) - fused[0]) <= 2e-6 * scale + 4e-5).all()
  _close(eng.local_energy()[0], fused[1], 4e-4)
  _check_forward_and_sampler(eng, theta, cfg, bonds, geom, ansatz, L, nonlin, b)
  eng.close()


def _check_gradients(eng, theta, cfg, bonds, geom, ansatz, L, nonlin, b):
  """Two accumulate calls (the second on moved chains) against the oracle's manual back-propagation, the gradient
  formula, one Adam step: tests/test_gpu_conv.py::test_conv_energy_gradient_accumulators on this path."""
  from cgs_vmc_amd import _hip
  acc = vo.Accumulators(theta.size, np.float64)
  eng.set_bonds(bonds, -1.0, 1.0)
  eng.reset_accumulators()
  vo.energy_gradient_accumulate(acc, theta, cfg, bonds, -1.0, 1.0, -10.0, geom, L, np.float64,
                                ansatz=ansatz, nonlinearity=nonlin)
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  eng.mc_steps(3)
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
  cancel = 4e-6 * max(np.abs(acc.g2_total).max(), abs(acc.mean_energy()) * np.abs(acc.g1_total).max()) / 2
  assert np.abs(grad - grad_ref).max() < 2e-3 * np.abs(grad_ref).max() + 2e-4 + cancel
  st = vo.AdamState(p)
  th_ref = vo.adam_apply(st, theta, grad, 1e-3, 0.9, 0.99, 1e-8)
  eng.apply_adam(_hip.VMC_MODE_ENERGY_GRADIENT, 1e-3, 0.9, 0.99, 1e-8)
  np.testing.assert_allclose(eng.get_params(), th_ref, rtol=0, atol=2e-6)
  _logits_close(eng.amplitude()[0], eng.get_params(), cur, ansatz, geom, L, nonlin)


GRAD_SHAPES = [s_ for s_ in GENERAL_SHAPES if s_[6] <= 100]      # (the 600-chain case: forward only; its oracle gradient is slow)


@pytest.mark.parametrize('ansatz,sx,sy,L,f,k,b,nonlin', GRAD_SHAPES, ids=['{}-{}x{}-L{}-F{}-K{}-B{}-{}'.format(*s_) for s_ in GRAD_SHAPES])
def test_general_convolution_gradient_accumulators(ansatz, sx, sy, L, f, k, b, nonlin):
  eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, nonlin)
  assert eng.kernel_path() == 6
  _check_gradients(eng, theta, cfg, bonds, geom, ansatz, L, nonlin, b)
  eng.close()


@pytest.mark.parametrize('ansatz,sx,sy,L,f,k,b,nonlin', BOTH)
def test_general_convolution_gradient_accumulators_at_fused_shapes(monkeypatch, ansatz, sx, sy, L, f, k, b, nonlin):
  monkeypatch.setenv('CGS_VMC_CONV_GENERAL', '1')
  eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, nonlin)
  assert eng.kernel_path() == 6
  _check_gradients(eng, theta, cfg, bonds, geom, ansatz, L, nonlin, b)
  eng.close()


@pytest.mark.parametrize('ansatz,sx,sy,L,f,k,b,nonlin,n_store,forced', [
    ('conv_2d', 4, 4, 2, 8, 3, 16, 'relu', 2, True),          # shapes of tests/test_gpu_sr.py, forced onto this path
    ('conv_2d', 4, 4, 2, 24, 3, 12, 'cos', 2, True),
    ('res_net_2d', 4, 4, 2, 8, 3, 16, 'relu', 2, True),
    ('conv_1d', 12, 1, 3, 12, 5, 14, 'sigmoid', 3, True),
    ('conv_2d', 7, 7, 2, 8, 7, 10, 'relu', 2, True),
    ('conv_2d', 4, 4, 2, 66, 1, 10, 'relu', 2, False),        # beyond the fused limits: 66 filters (1 x 1 taps: 4,554 parameters)
    ('conv_1d', 12, 1, 2, 4, 11, 9, 'tanh', 2, False),        # ... an 11-tap kernel
    ('res_net_1d', 14, 1, 1, 6, 10, 8, 'relu', 2, False),
])
def test_general_convolution_stochastic_reconfiguration(monkeypatch, ansatz, sx, sy, L, f, k, b, nonlin, n_store, forced):
  """SR (an extension) on the general path, single-rank solves: only the chains are stored, every CG iteration re-runs
  the taped forward and the backward of the stored chains, the per-sample weights are centred.  The checks are
  tests/test_gpu_sr.py's, at its bounds (explicit S from the oracle's per-sample gradients, matvec, solution through its
  fp64 residual and through O_c x)."""
  from tests.test_gpu_sr import test_sr_convolutional_matvec_and_solution as check
  if forced:
    monkeypatch.setenv('CGS_VMC_CONV_GENERAL', '1')
  from cgs_vmc_amd.engine import VmcEngine
  eng = VmcEngine(sx * sy, b, L, f, nonlinearity=nonlin, ansatz=ansatz, kernel_size=k, size_x=sx, size_y=sy)
  assert eng.kernel_path() == 6
  eng.close()
  check(ansatz, sx, sy, L, f, k, b, nonlin, n_store)


RING_SHAPES = [
    # the implicit-gather ring GEMM (k_gemm_ring<., true>: filters a multiple of 32, forced wherever it applies with
    # CGS_VMC_GEMM128=5) beyond the one 128-filter conv_2d case above: residual-block epilogues (selu / accumulate)
    # on the ring, the 1-D wrap, an even kernel's asymmetric padding, the narrow 64- and 96-column tiles, ragged last row tiles
    ('res_net_2d', 6, 6, 1, 64, 3, 24, 'relu'),
    ('conv_1d', 40, 1, 3, 96, 5, 12, 'relu'),
    ('conv_2d', 8, 6, 2, 64, 4, 10, 'tanh'),
    ('conv_2d', 6, 6, 2, 96, 3, 14, 'relu'),
]


@pytest.mark.parametrize('ansatz,sx,sy,L,f,k,b,nonlin', RING_SHAPES,
                         ids=['{}-{}x{}-L{}-F{}-K{}-B{}-{}'.format(*s) for s in RING_SHAPES])
def test_implicit_gather_ring_agrees_with_explicit_im2col(monkeypatch, ansatz, sx, sy, L, f, k, b, nonlin):
  """CGS_VMC_CONV_GENERAL_IMPLICIT (read per call since round 6) = 1 / 0 in ONE engine: the gather inside the ring GEMM's A
  operand against the im2col matrix + the same GEMM -- logits, local energies and gradient sums agree to the
  summation order, and the implicit results meet the oracle's bars."""
  from cgs_vmc_amd import _hip
  monkeypatch.setenv('CGS_VMC_GEMM128', '5')
  monkeypatch.setenv('CGS_VMC_CONV_GENERAL', '1')        # (64 filters are within the fused kernels' limits)
  eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, nonlin)
  assert eng.kernel_path() == 6
  out = {}
  for implicit in ('0', '1'):
    monkeypatch.setenv('CGS_VMC_CONV_GENERAL_IMPLICIT', implicit)
    eng.set_configs(cfg)
    logit = eng.amplitude(cfg)[0]
    eloc = eng.local_energy()[0]
    eng.reset_accumulators()
    eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
    out[implicit] = (logit, eloc, eng.get_accumulators())
  _, scale = vo.conv_forward(theta, cfg, ansatz, geom, L, nonlin, np.float64, return_tape='scale')
  tol = 1e-6 * scale + 2e-5
  _logits_close(out['1'][0], theta, cfg, ansatz, geom, L, nonlin)
  _logits_close(out['0'][0], theta, cfg, ansatz, geom, L, nonlin)
  assert (np.abs(out['1'][0].astype(np.float64) - out['0'][0]) <= 2 * tol).all()
  assert np.abs(out['1'][1] - out['0'][1]).max() <= 4e-4 * max(1.0, np.abs(out['0'][1]).max())
  assert np.abs(out['1'][2] - out['0'][2]).max() <= 4e-3 * np.abs(out['0'][2]).max() + 2e-4
  psi_fn = vo.ANSATZ[ansatz][0]
  amp = lambda c: psi_fn(theta, c, geom, L, nonlinearity=nonlin, dtype=np.float64)
  _close(out['1'][1], vo.local_value(amp, cfg, bonds, -1.0, 1.0, dtype=np.float64), 2e-4)
  eng.close()


BAND_SHAPES = [
    # ansatz, size_x, size_y, layers / blocks, filters, kernel, B, nonlinearity: <= 16 filters, 2 .. 7 taps per axis
    ('conv_2d', 36, 36, 3, 16, 5, 3, 'relu'),       # maps beyond the LDS of the fused kernels: three bands of 12 lattice rows
    ('conv_2d', 20, 12, 2, 5, 4, 7, 'tanh'),        # 5 filters (Fp = 8), even kernel, ragged last tile
    ('conv_2d', 3, 4, 2, 16, 7, 9, 'relu'),         # lattice smaller than the kernel: the wrap runs around more than once
    ('conv_2d', 40, 9, 2, 12, 3, 5, 'cos'),         # pre-activation maps (the cosine is applied while the band is staged)
    ('res_net_2d', 24, 24, 2, 16, 5, 4, 'relu'),    # selu / residual-add epilogues
    ('conv_1d', 300, 1, 3, 16, 7, 6, 'sigmoid'),    # 1-D: K x 1 taps
    ('res_net_1d', 64, 1, 1, 9, 2, 8, 'relu'),
    # more than 16 filters: one workgroup per output channel block, the band block-major in LDS
    ('conv_2d', 20, 20, 2, 64, 3, 4, 'relu'),       # four blocks, 3 x 3: 144 fragments per output block; maps beyond the LDS
    ('conv_2d', 16, 16, 2, 32, 5, 4, 'relu'),       # two blocks, 5 x 5: 200 fragments
    ('conv_2d', 8, 6, 2, 48, 4, 10, 'tanh'),        # three blocks, even kernel
    ('res_net_2d', 6, 6, 1, 40, 3, 12, 'relu'),     # 40 filters: the third block half empty; residual epilogues
    ('conv_1d', 50, 1, 2, 64, 7, 6, 'relu'),        # 1-D, four blocks x 7 taps
]


@pytest.mark.parametrize('ansatz,sx,sy,L,f,k,b,nonlin', BAND_SHAPES,
                         ids=['{}-{}x{}-L{}-F{}-K{}-B{}-{}'.format(*s) for s in BAND_SHAPES])
def test_band_kernel_agrees_with_im2col_gemm_and_the_oracle(monkeypatch, ansatz, sx, sy, L, f, k, b, nonlin):
  """Round 6: at <= 64 filters (while one output block's fragments fit the registers) the general path's convolutions run on k_cgen_band (conv_band.hip: bands of lattice rows
  staged through LDS, no im2col matrix).  Same engine, CGS_VMC_CONV_BAND=0 / 1 (read per call): logits, local energies,
  a sampler trajectory and the gradient sums (whose taped forward takes the band kernel too) agree to fp32 summation-order
  differences, and the band results meet tests/test_gpu_conv.py's bars against the fp64 oracle."""
  monkeypatch.setenv('CGS_VMC_CONV_GENERAL', '1')
  from cgs_vmc_amd import _hip
  eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, nonlin)
  assert eng.kernel_path() == 6
  out = {}
  for band in ('0', '1'):
    monkeypatch.setenv('CGS_VMC_CONV_BAND', band)
    eng.set_configs(cfg)
    eng.step_counter = 0
    logit = eng.amplitude(cfg)[0]
    eloc = eng.local_energy()[0]
    eng.reset_accumulators()
    eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
    acc = eng.get_accumulators()
    eng.mc_steps(5)
    out[band] = (logit, eloc, acc, eng.get_configs(), eng.amplitude()[0])
  _, scale = vo.conv_forward(theta, cfg, ansatz, geom, L, nonlin, np.float64, return_tape='scale')
  tol = 1e-6 * scale + 2e-5                            # tests/test_gpu_conv.py::_logits_close's bound against the oracle
  _logits_close(out['1'][0], theta, cfg, ansatz, geom, L, nonlin)
  _logits_close(out['0'][0], theta, cfg, ansatz, geom, L, nonlin)
  assert (np.abs(out['1'][0].astype(np.float64) - out['0'][0]) <= 2 * tol).all()
  assert np.abs(out['1'][1] - out['0'][1]).max() <= 4e-4 * max(1.0, np.abs(out['0'][1]).max())
  assert np.abs(out['1'][2] - out['0'][2]).max() <= 4e-3 * np.abs(out['0'][2]).max() + 2e-4
  same = (out['1'][3] == out['0'][3]).all(1)          # chains whose five accept tests fell the same way
  assert same.mean() >= 0.8
  _logits_close(out['1'][4][same], theta, out['1'][3][same], ansatz, geom, L, nonlin)
  eng.close()


@pytest.mark.parametrize('ansatz,sx,sy,L,f,k,b,nonlin', [('conv_2d', 12, 12, 2, 16, 5, 9, 'relu'),       # band kernel
                                                         ('res_net_2d', 4, 4, 2, 96, 3, 8, 'relu'),      # GEMM form, residual blocks
                                                         ('conv_1d', 30, 1, 3, 100, 5, 9, 'relu')])
def test_step_tail_launch_gives_the_chains_of_the_four_launches(monkeypatch, ansatz, sx, sy, L, f, k, b, nonlin):
  """Round 6: k_cgen_step_tail (map sum + candidate logit + accept + next proposal in one launch) against the four
  launches it replaces (CGS_VMC_CONV_STEP_TAIL=0, read per call): the same arithmetic in the same order, so chains,
  accept counts and cached logits are the same bits, sweep after sweep."""
  monkeypatch.setenv('CGS_VMC_CONV_GENERAL', '1')
  eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, nonlin)
  assert eng.kernel_path() == 6
  out = {}
  for tail in ('0', '1'):
    monkeypatch.setenv('CGS_VMC_CONV_STEP_TAIL', tail)
    eng.set_configs(cfg)
    eng.step_counter = 0
    acc1 = eng.mc_steps(7)
    c1 = eng.get_configs()
    acc2 = eng.mc_steps(2 * sx * sy)
    out[tail] = (acc1, c1, acc2, eng.get_configs(), eng.amplitude()[0])
  assert 0 < out['1'][2] <= 2 * sx * sy * b
  for a, bb in zip(out['0'], out['1']):
    np.testing.assert_array_equal(a, bb)
  eng.close()


@pytest.mark.parametrize('ansatz,sx,sy,L,f,k,b,nonlin', [('conv_2d', 12, 12, 2, 16, 5, 9, 'relu'),       # band kernel
                                                         ('conv_2d', 10, 10, 3, 128, 3, 37, 'relu'),     # implicit-gather ring, 29 row tiles
                                                         ('res_net_2d', 4, 4, 2, 96, 3, 8, 'relu'),      # residual blocks
                                                         ('conv_1d', 30, 1, 3, 100, 5, 3, 'cos')])       # im2col form; 3 chains in 4 groups
def test_sampler_chain_groups_leave_the_chains_alone(monkeypatch, ansatz, sx, sy, L, f, k, b, nonlin):
  """Round 6: the general sampler runs its batch as G chain groups on streams of their own (run_sweep_cgen; default 2,
  CGS_VMC_CONV_GENERAL_GROUPS read per call).  A row's value does not depend on the launch or tile it sits in, so chains,
  accept counts and cached logits are the same bits for every G -- and the groups' maps are disjoint slices."""
  monkeypatch.setenv('CGS_VMC_CONV_GENERAL', '1')
  eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, nonlin)
  assert eng.kernel_path() == 6
  out = {}
  for groups in ('1', '2', '3', '4'):
    monkeypatch.setenv('CGS_VMC_CONV_GENERAL_GROUPS', groups)
    eng.set_configs(cfg)
    eng.step_counter = 0
    acc1 = eng.mc_steps(5)
    c1 = eng.get_configs()
    acc2 = eng.mc_steps(sx * sy)
    out[groups] = (acc1, c1, acc2, eng.get_configs(), eng.amplitude()[0])
  assert 0 < out['1'][2] <= sx * sy * b
  for groups in ('2', '3', '4'):
    for a, bb in zip(out['1'], out[groups]):
      np.testing.assert_array_equal(a, bb)
  eng.close()


PATCH_SHAPES = [
    ('conv_2d', 20, 20, 3, 16, 3, 5, 'relu'),       # boxes of 3, 5, 7 sites per axis
    ('conv_2d', 36, 36, 3, 16, 5, 3, 'relu'),       # bench workload heisenberg36x36_conv3x16k5_b32: 5, 9, 13
    ('conv_2d', 14, 18, 2, 12, 4, 4, 'tanh'),       # even kernel (1 in front, 2 behind), 12 filters, a non-square lattice
    ('conv_2d', 16, 16, 2, 8, 5, 4, 'cos'),         # the cosine: the maps hold pre-activations, applied on the gather
    ('conv_2d', 7, 9, 3, 16, 3, 4, 'relu'),         # the last box as wide as the lattice along one axis
    ('conv_2d', 24, 24, 4, 16, 2, 3, 'sigmoid'),    # four convolutions of 2 x 2 taps
    ('conv_1d', 40, 1, 3, 16, 5, 4, 'relu'),        # 1-D: boxes of 5, 9, 13 sites
    ('conv_1d', 30, 1, 2, 10, 6, 5, 'tanh'),        # ... even kernel: 3 in front, 2 behind
    ('res_net_2d', 20, 20, 1, 16, 3, 4, 'relu'),    # one residual block: initial convolution, selu, h + second convolution
    ('res_net_2d', 24, 22, 2, 12, 3, 3, 'relu'),    # two blocks: five convolutions, boxes up to 11 sites
    ('res_net_1d', 60, 1, 2, 16, 5, 4, 'relu'),     # 1-D residual network: boxes up to 21 sites
]


@pytest.mark.parametrize('ansatz,sx,sy,L,f,k,b,nonlin', PATCH_SHAPES, ids=['{}-{}x{}-L{}-F{}-K{}-B{}-{}'.format(*s) for s in PATCH_SHAPES])
def test_patch_sampler_gives_the_chains_of_the_full_forward(monkeypatch, ansatz, sx, sy, L, f, k, b, nonlin):
  """Round 6: k_cgen_patch_sweep (conv_patch.hip) recomputes, per step, the two boxes of every convolution that the
  exchanged pair reaches -- with the band kernel's tile arithmetic and the step tail's sum order -- instead of the whole
  lattice.  Chains, accept counts and cached logits are the bits of the full-forward sampler (CGS_VMC_CONV_PATCH=0, read
  per call), over launches long enough for accepted moves to be carried from step to step through the stored maps."""
  monkeypatch.setenv('CGS_VMC_CONV_GENERAL', '1')
  eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, nonlin)
  assert eng.kernel_path() == 6
  n = sx * sy
  out = {}
  for patch in ('0', '2'):
    monkeypatch.setenv('CGS_VMC_CONV_PATCH', patch)
    eng.set_configs(cfg)
    eng.step_counter = 0
    acc1 = eng.mc_steps(3)
    c1 = eng.get_configs()
    acc2 = eng.mc_steps(2 * n)
    c2 = eng.get_configs()
    l2 = eng.amplitude()[0]
    acc3 = eng.mc_steps(n // 2)
    out[patch] = (acc1, c1, acc2, c2, l2, acc3, eng.get_configs(), eng.amplitude()[0])
  assert 0 < out['0'][2] <= 2 * n * b
  for a, bb in zip(out['0'], out['2']):
    np.testing.assert_array_equal(a, bb)
  _logits_close(out['2'][7], theta, out['2'][6], ansatz, geom, L, nonlin)      # ... and the oracle's amplitudes of the final chains
  eng.close()


@pytest.mark.parametrize('ansatz,sx,sy,L,f,k,b,nonlin', PATCH_SHAPES, ids=['{}-{}x{}-L{}-F{}-K{}-B{}-{}'.format(*s) for s in PATCH_SHAPES])
def test_patch_rows_give_the_local_energies_of_the_full_forward(monkeypatch, ansatz, sx, sy, L, f, k, b, nonlin):
  """Round 6: the local energies' connected configurations (operators.py:162-169: the chain with one antiparallel bond
  exchanged) through the ELOC form of k_cgen_patch_sweep -- the chains' maps once, per row the boxes around the bond's two
  sites, the last map's sum with them overlaid in k_cgen_rowsum's order -- against a full forward of every row
  (CGS_VMC_CONV_PATCH=0): local energies, their terms and the gradient accumulators are the same bits; and the oracle's
  local energies within the path's tolerance."""
  monkeypatch.setenv('CGS_VMC_CONV_GENERAL', '1')
  eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, nonlin)
  assert eng.kernel_path() == 6
  out = {}
  for jx in (-1.0, 1.0):
    eng.set_bonds(bonds, jx, 1.0)
    for patch in ('0', '2'):
      monkeypatch.setenv('CGS_VMC_CONV_PATCH', patch)
      eng.set_configs(cfg)
      eloc, mean = eng.local_energy()
      diag, off = eng.local_energy_terms()
      eng.reset_accumulators()
      eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
      out[patch] = (eloc, np.float64(mean), diag, off, eng.get_accumulators())
    assert eng.last_connected_rows() > b
    for a, bb in zip(out['0'], out['2']):
      np.testing.assert_array_equal(a, bb)
    amp = lambda c: vo.ANSATZ[ansatz][0](theta, c, geom, L, nonlinearity=nonlin, dtype=np.float64)
    if sx * sy <= 1000:      # (the oracle works on amplitudes: at 36 x 36 sites exp(logit) leaves the doubles -- the bits above are the check there)
      _close(out['2'][0], vo.local_value(amp, cfg, bonds, jx, 1.0, dtype=np.float64), 2e-4)
  eng.close()


def test_wide_lattice_goes_to_the_patch_kernels_by_itself(monkeypatch):
  """plan_desc sends a shape the fused kernels would take to the general path where the patch kernels beat them (the boxes at
  most a fifth of a forward: plan_cgen_patch_routes); CGS_VMC_CONV_GENERAL=0 keeps the fused kernels.  The two agree as the
  two paths do everywhere, and the routed ctx passes the oracle checks."""
  ansatz, sx, sy, L, f, k, b, nonlin = 'conv_2d', 20, 20, 3, 16, 3, 6, 'relu'
  monkeypatch.setenv('CGS_VMC_CONV_GENERAL', '0')
  eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, nonlin)
  assert eng.kernel_path() == 3 and not eng.conv_patch(sx * sy)
  fused = (eng.amplitude()[0], eng.local_energy()[0])
  eng.close()
  monkeypatch.delenv('CGS_VMC_CONV_GENERAL')
  monkeypatch.delenv('CGS_VMC_CONV_PATCH', raising=False)
  eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, nonlin)
  assert eng.kernel_path() == 6 and eng.conv_patch(sx * sy) and not eng.conv_patch(4)
  _, scale = vo.conv_forward(theta, cfg, ansatz, geom, L, nonlin, np.float64, return_tape='scale')
  assert (np.abs(eng.amplitude()[0].astype(np.float64) - fused[0]) <= 2e-6 * scale + 4e-5).all()
  _close(eng.local_energy()[0], fused[1], 4e-4)
  _check_forward_and_sampler(eng, theta, cfg, bonds, geom, ansatz, L, nonlin, b, steps=10)
  eng.close()
  # a lattice the network's reach covers stays where it was
  eng, *_ = _make('conv_2d', 12, 12, 3, 16, 3, 6, 'relu')
  assert eng.kernel_path() == 3
  eng.close()


def test_patch_sampler_against_the_oracle_trajectory(monkeypatch):
  """The oracle checks of every general-path shape (amplitudes, local energies, proposals, injected steps, trajectory)
  with the patch sampler forced on for the plain launches."""
  monkeypatch.setenv('CGS_VMC_CONV_GENERAL', '1')
  monkeypatch.setenv('CGS_VMC_CONV_PATCH', '2')
  ansatz, sx, sy, L, f, k, b, nonlin = 'conv_2d', 18, 18, 3, 16, 3, 6, 'relu'
  eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, nonlin)
  assert eng.kernel_path() == 6
  _check_forward_and_sampler(eng, theta, cfg, bonds, geom, ansatz, L, nonlin, b, steps=12)
  eng.close()


def test_general_convolution_sr_keeps_the_tape_across_cg_iterations(monkeypatch):
  """Round 6: with the stored chains in one block the taped forward and the backward run once per solve instead of once
  per CG iteration (CGS_VMC_SR_KEEP_TAPE=0: every iteration, read per call) -- the same numbers go into every product,
  so iterations, residual and solution are the same bits; and the solve is faster."""
  import time
  from cgs_vmc_amd.engine import VmcEngine
  ansatz, n, L, f, k, b, n_store = 'conv_1d', 24, 3, 8, 11, 64, 3
  geom = (f, k, n, 1)
  rng = np.random.default_rng(0)
  theta = vo.conv_init_params(ansatz, geom, L, rng)
  theta += (0.03 * rng.standard_normal(theta.size)).astype(np.float32)
  eng = VmcEngine(n, b, L, f, nonlinearity='tanh', seed=2024, ansatz=ansatz, kernel_size=k, size_x=n, size_y=1)
  assert eng.kernel_path() == 6
  eng.set_params(theta)
  eng.set_bonds(vo.chain_bonds(n), -1.0, 1.0)
  eng.sr_reserve(n_store)
  eng.reset_accumulators()
  for j in range(n_store):
    eng.set_configs(vo.random_configurations(n, b, np.random.RandomState(20 + j)))
    eng.accumulate(0)
  out, secs = {}, {}
  for keep in ('0', '1', '0', '1'):
    monkeypatch.setenv('CGS_VMC_SR_KEEP_TAPE', keep)
    eng.synchronize()
    t0 = time.perf_counter()
    it, res = eng.sr_solve(0.01, 0.0, 40)
    x = eng.sr_get_solution()
    secs[keep] = time.perf_counter() - t0
    out[keep] = (it, res, x)
  assert out['0'][0] == out['1'][0] == 40 and out['0'][1] == out['1'][1]
  np.testing.assert_array_equal(out['0'][2], out['1'][2])
  print('\n40 CG iterations over {} stored chains: {:.1f} ms recomputing the tape, {:.1f} ms keeping it'.format(
      n_store * b, 1e3 * secs['0'], 1e3 * secs['1']))
  assert secs['1'] < secs['0']
  eng.close()


def test_general_convolution_sr_op_by_op_two_phase_matvec():
  """The op-by-op CG loop on the general path (round 6): vmc_sr_matvec_phase1 -> [all-reduce of the buffer's last float]
  -> vmc_sr_matvec_phase2 -> [all-reduce of the buffer] -> vmc_sr_cg_update arrives where the one-call vmc_sr_solve
  does; vmc_sr_matvec_partial still refuses (it cannot know the mean over all ranks) and says what to call."""
  from cgs_vmc_amd.engine import VmcEngine
  ansatz, n, L, f, k, b, n_store = 'conv_1d', 12, 2, 4, 11, 18, 2
  geom = (f, k, n, 1)
  rng = np.random.default_rng(0)
  theta = vo.conv_init_params(ansatz, geom, L, rng)
  theta += (0.03 * rng.standard_normal(theta.size)).astype(np.float32)
  eng = VmcEngine(n, b, L, f, nonlinearity='tanh', seed=2024, ansatz=ansatz, kernel_size=k, size_x=n, size_y=1)
  assert eng.kernel_path() == 6
  eng.set_params(theta)
  eng.set_bonds(vo.chain_bonds(n), -1.0, 1.0)
  eng.sr_reserve(n_store)
  eng.reset_accumulators()
  for j in range(n_store):
    eng.set_configs(vo.random_configurations(n, b, np.random.RandomState(20 + j)))
    eng.accumulate(0)
  lam, tol = 0.01, 1e-6
  it_ref, res_ref = eng.sr_solve(lam, tol, 500)
  x_ref = eng.sr_get_solution()
  rr0 = rr = eng.sr_begin()
  with pytest.raises(Exception, match='vmc_sr_matvec_phase1'):
    eng.sr_matvec_partial()
  with pytest.raises(Exception, match='phase1 first'):
    eng.sr_matvec_phase2()
  it = 0
  while it < 500 and rr > tol * tol * rr0:
    eng.sr_matvec_phase1()
    buf = eng.sr_get_buffer()
    assert not buf[:-1].any()                  # only the last float (sum_b O_b . p) travels between the phases
    eng.sr_matvec_phase2()
    rr = eng.sr_cg_update(lam)
    it += 1
  x = eng.sr_get_solution()
  assert abs(it - it_ref) <= 2, (it, it_ref)
  assert np.abs(x - x_ref).max() <= 1e-4 * np.abs(x_ref).max(), (it, it_ref)
  eng.close()


def test_general_convolution_through_run_training_and_evaluation(tmp_path):
  """--wavefunction_type=conv_1d with kernel_size=11 (beyond the fused kernels: the general path) through the
  run_training / run_energy_evaluation counterparts on the reference's default lattice, the periodic chain
  (run_training.py:103-109), 16 sites: the variational energy approaches E0 = -7.1423 (exact diagonalisation) from
  above.  (A test of this kind with MORE THAN 64 FILTERS does not exist for a reason that is the ansatz's, not the
  path's: the logit is the plain sum of N x filters outputs -- wavefunctions.py:569 -- whose spread at the Sonnet
  initialisation grows with the filter count, ~ +-25 at 72 filters on 16 sites, so psi'/psi = exp(+-25) and the first
  Adam steps diverge at any learning rate the reference's schedule offers.)"""
  import os
  from cgs_vmc_amd import run_energy_evaluation, run_training, session as session_lib, wavefunctions
  session_lib.reset_default_graph()
  wavefunctions.reset_name_scope()
  os.environ.update(CGS_VMC_SEED='77', CGS_VMC_CONFIG_SEED='5', CGS_VMC_INIT_SEED='31')
  d = str(tmp_path)
  hp = ('batch_size=256,num_conv_layers=2,num_conv_filters=8,kernel_size=11,num_equilibration_sweeps=10,'
        'num_batches_per_epoch=8,learning_rates=[0.003,0.001],learning_rate_stops=[200]')
  run_training.main(['--checkpoint_dir', d, '--num_sites', '16', '--heisenberg_jx', '-1.0',
                     '--wavefunction_type', 'conv_1d', '--optimizer', 'EnergyGradient',
                     '--num_epochs', '250', '--hparams', hp])
  energies = [float(x) for x in open(os.path.join(d, 'metrics.txt')).read().split()]
  tail = np.mean(energies[-10:])
  assert -7.1423 - 0.05 < tail < -6.0, (tail, energies[::25])
  session_lib.reset_default_graph()
  wavefunctions.reset_name_scope()
  run_energy_evaluation.main(['--checkpoint_dir', d, '--heisenberg_jx', '-1.0', '--hparams', 'num_evaluation_samples=5'])


def test_patch_kernels_through_run_training(tmp_path, monkeypatch):
  """--wavefunction_type=conv_1d on a 40-site chain, two convolutions of 3 taps: plan_desc routes the shape to the general
  path's patch kernels.  Six EnergyGradient epochs through the run_training counterpart, with the patch kernels and with
  full forwards of every candidate and connected configuration (CGS_VMC_CONV_PATCH=0): chains, local energies and gradient
  sums are the same bits, so the two energy histories are the same numbers."""
  import os
  from cgs_vmc_amd import run_training, session as session_lib, wavefunctions
  hp = ('batch_size=64,num_conv_layers=2,num_conv_filters=8,kernel_size=3,num_equilibration_sweeps=2,'
        'num_batches_per_epoch=4,learning_rates=[0.002],learning_rate_stops=[]')
  from cgs_vmc_amd.engine import VmcEngine
  monkeypatch.delenv('CGS_VMC_CONV_GENERAL', raising=False)
  probe = VmcEngine(40, 64, 2, 8, ansatz='conv_1d', kernel_size=3)
  assert probe.kernel_path() == 6 and probe.conv_patch(40)      # (the shape this training runs at)
  probe.close()
  histories = []
  for patch in ('1', '0'):
    monkeypatch.setenv('CGS_VMC_CONV_PATCH', patch)
    session_lib.reset_default_graph()
    wavefunctions.reset_name_scope()
    os.environ.update(CGS_VMC_SEED='7', CGS_VMC_CONFIG_SEED='5', CGS_VMC_INIT_SEED='3')
    d = str(tmp_path / ('patch' + patch))
    os.makedirs(d)
    run_training.main(['--checkpoint_dir', d, '--num_sites', '40', '--heisenberg_jx', '-1.0',
                       '--wavefunction_type', 'conv_1d', '--optimizer', 'EnergyGradient',
                       '--num_epochs', '6', '--hparams', hp])
    histories.append([float(x) for x in open(os.path.join(d, 'metrics.txt')).read().split()])
  assert len(histories[0]) == 6 and np.isfinite(histories[0]).all()
  assert histories[0] == histories[1], histories
  session_lib.reset_default_graph()
  wavefunctions.reset_name_scope()


@pytest.mark.parametrize('ansatz,f,k', [('conv_2d', 8, 3), ('res_net_2d', 8, 3), ('conv_2d', 80, 3), ('conv_1d', 6, 11)])
def test_general_convolution_log_overlap_itswo_accumulators(monkeypatch, ansatz, f, k):
  """training.py:655-705 on the general path (forced where the fused kernels would take the shape): the supervisor's
  amplitudes (parameter set 1), the overlap ratio and the weighted gradient sums against the oracle."""
  from cgs_vmc_amd import _hip
  monkeypatch.setenv('CGS_VMC_CONV_GENERAL', '1')
  sx, sy, L, b = (16, 1, 2, 48) if ansatz == 'conv_1d' else (4, 4, 2, 48)
  eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, 'relu')
  assert eng.kernel_path() == 6
  eng.transfer_params()
  rng = np.random.default_rng(8)
  theta2 = theta + (0.02 / np.sqrt(max(1.0, f / 8.0)) * rng.standard_normal(theta.size)).astype(np.float32)
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
def test_general_convolution_non_exp_output_activation(monkeypatch, oact):
  """wavefunctions.py:576-579 on the general path: psi = g(sum), linear-domain ratios, O_k with g'(x) / g(x)."""
  from cgs_vmc_amd import _hip
  monkeypatch.setenv('CGS_VMC_CONV_GENERAL', '1')
  ansatz, sx, sy, L, f, k, b = 'conv_2d', 4, 4, 2, 8, 3, 32
  eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, 'tanh', output_activation=oact, noise=0.01)
  assert eng.kernel_path() == 6
  amp = lambda c: vo.ANSATZ[ansatz][0](theta, c, geom, L, nonlinearity='tanh', output_activation=oact, dtype=np.float64)
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


@pytest.mark.parametrize('ansatz,sx,sy,L,f,k,patch', [('conv_2d', 6, 6, 2, 72, 3, False),
                                                      ('conv_2d', 20, 20, 3, 16, 3, True)])     # routed to the patch kernels
def test_general_convolution_shard_invariance_and_reproducibility(ansatz, sx, sy, L, f, k, patch):
  """Chains [16, 48) of a 64-chain run walk the same trajectory as a 32-chain shard with chain_offset 16 (Philox keyed
  by the global chain id; no float atomics, a fixed order of additions: the energies are the same bits too), two
  identical runs agree bit for bit -- and so do the gradient sums of two identical accumulate calls (split-K partials
  folded in slice order).  Also on the patch kernels (a chain is a workgroup's business; a row's boxes do not depend on
  the rows beside it)."""
  from cgs_vmc_amd.engine import VmcEngine
  n, geom = sx * sy, (f, k, sx, sy)
  rng = np.random.default_rng(3)
  theta = (0.5 * vo.conv_init_params(ansatz, geom, L, rng)).astype(np.float32)
  cfg = vo.random_configurations(n, 64, np.random.RandomState(4))
  outs = []
  for (b, off, rows) in ((64, 0, slice(0, 64)), (64, 0, slice(0, 64)), (32, 16, slice(16, 48))):
    eng = VmcEngine(n, b, L, f, seed=11, ansatz=ansatz, kernel_size=k, size_x=sx, size_y=sy, chain_offset=off)
    assert eng.kernel_path() == 6 and eng.conv_patch(n) == patch
    eng.set_params(theta); eng.set_configs(cfg[rows]); eng.set_bonds(vo.torus_bonds(sx, sy), -1.0, 1.0)
    eng.mc_steps(n)
    eng.reset_accumulators()
    eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
    outs.append((eng.get_configs(), eng.local_energy()[0], eng.get_accumulators()))
    eng.close()
  np.testing.assert_array_equal(outs[0][0], outs[1][0])
  np.testing.assert_array_equal(outs[0][1], outs[1][1])
  np.testing.assert_array_equal(outs[0][2], outs[1][2])
  np.testing.assert_array_equal(outs[0][0][16:48], outs[2][0])
  np.testing.assert_array_equal(outs[0][1][16:48], outs[2][1])


def _random_general_shapes(count, seed=5055):
  """Seeded random geometries BEYOND the fused limits: 65 .. 160 filters (every residue mod 4 and mod 32) or a
  10 .. 13-tap kernel; every padding parity, ragged batches, kernels longer than a lattice side."""
  rng = np.random.default_rng(seed)
  acts = ['relu', 'tanh', 'sigmoid', 'identity', 'cos']
  shapes = []
  while len(shapes) < count:
    ansatz = ['conv_2d', 'res_net_2d', 'conv_1d', 'res_net_1d'][int(rng.integers(4))]
    wide = len(shapes) % 2 == 0
    k = int(rng.integers(1, 6)) if wide else int(rng.integers(10, 14))
    f = int(rng.integers(65, 161)) if wide else int(rng.integers(1, 13))
    if ansatz in vo.CONV_1D:
      sx, sy = int(rng.integers(max(2, (k + 1) // 2), 29)), 1
    else:
      sx, sy = int(rng.integers(max(2, (k + 1) // 2), 9)), int(rng.integers(max(2, (k + 1) // 2), 9))
    if (sx * sy) % 2 or sx * sy < 4:
      continue
    resnet = ansatz.startswith('res_net')
    L = int(rng.integers(0, 3)) if resnet else int(rng.integers(1, 4))
    b = int(rng.integers(1, 25))
    nonlin = 'relu' if resnet else acts[int(rng.integers(len(acts)))]
    shapes.append((ansatz, sx, sy, L, f, k, b, nonlin))
  return shapes


RANDOM_GENERAL = _random_general_shapes(20)


@pytest.mark.parametrize('ansatz,sx,sy,L,f,k,b,nonlin', RANDOM_GENERAL,
                         ids=['{}-{}x{}-L{}-F{}-K{}-B{}-{}'.format(*s_) for s_ in RANDOM_GENERAL])
def test_general_convolution_random_shapes(ansatz, sx, sy, L, f, k, b, nonlin):
  """tests/test_gpu_conv.py::test_conv_random_shapes beyond the fused limits: amplitudes, local energies, the gradient
  sums and one injected mc_step on random geometries."""
  from cgs_vmc_amd import _hip
  eng, theta, cfg, bonds, geom = _make(ansatz, sx, sy, L, f, k, b, nonlin, seed=b + 7 * k)
  assert eng.kernel_path() == 6
  n = sx * sy
  psi_fn = vo.ANSATZ[ansatz][0]
  amp = lambda c: psi_fn(theta, c, geom, L, nonlinearity=nonlin, dtype=np.float64)
  _logits_close(eng.amplitude()[0], theta, cfg, ansatz, geom, L, nonlin)
  e_ref = vo.local_value(amp, cfg, bonds, -1.0, 1.0, dtype=np.float64)
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


@pytest.mark.parametrize('block_rows', [5, 16])
def test_general_convolution_in_several_blocks(monkeypatch, block_rows):
  """The block loops of the path -- row configurations are processed cg_rows at a time (sized for 768 MB of im2col rows:
  never more than one block at test sizes) -- forced to 5 / 16 rows per block: local energies over ragged last blocks, the
  gradient sums block by block, the two-pass SR matvec (the mean of O_b . v needs every block's t first)."""
  from tests.test_gpu_sr import test_sr_convolutional_matvec_and_solution as check_sr
  monkeypatch.setenv('CGS_VMC_CONV_GENERAL_BLOCK_ROWS', str(block_rows))
  for shape in (('conv_2d', 4, 4, 2, 80, 3, 12, 'relu'), ('res_net_2d', 4, 6, 1, 72, 3, 9, 'relu'), ('conv_1d', 26, 1, 2, 10, 12, 8, 'tanh')):
    eng, theta, cfg, bonds, geom = _make(*shape)
    assert eng.kernel_path() == 6
    _check_forward_and_sampler(eng, theta, cfg, bonds, geom, shape[0], shape[3], shape[7], shape[6], steps=3)
    eng.set_configs(cfg)
    _check_gradients(eng, theta, cfg, bonds, geom, shape[0], shape[3], shape[7], shape[6])
    eng.close()
  check_sr('conv_2d', 4, 4, 2, 66, 1, 10, 'relu', 2)
  check_sr('conv_1d', 12, 1, 2, 4, 11, 9, 'tanh', 2)


def test_general_convolution_stochastic_reconfiguration_through_run_training(tmp_path):
  """--optimizer StochasticReconfiguration (the extension's CLI name) with an 11-tap conv_1d network, i.e. on the general
  path, one rank: 16-site chain, E0 = -7.1423; SR descends much faster per epoch than Adam on the plain gradient."""
  import os
  from cgs_vmc_amd import run_training, session as session_lib, wavefunctions
  session_lib.reset_default_graph()
  wavefunctions.reset_name_scope()
  os.environ.update(CGS_VMC_SEED='77', CGS_VMC_CONFIG_SEED='5', CGS_VMC_INIT_SEED='31')
  d = str(tmp_path)
  hp = ('batch_size=256,num_conv_layers=2,num_conv_filters=8,kernel_size=11,num_equilibration_sweeps=10,'
        'num_batches_per_epoch=8,learning_rates=[0.05,0.02],learning_rate_stops=[30],'
        'sr_diag_shift=0.01,sr_cg_tolerance=0.001,sr_cg_max_iterations=200')
  run_training.main(['--checkpoint_dir', d, '--num_sites', '16', '--heisenberg_jx', '-1.0',
                     '--wavefunction_type', 'conv_1d', '--optimizer', 'StochasticReconfiguration',
                     '--num_epochs', '40', '--hparams', hp])
  energies = [float(x) for x in open(os.path.join(d, 'metrics.txt')).read().split()]
  tail = np.mean(energies[-5:])
  assert -7.1423 - 0.05 < tail < -6.5, (tail, energies[::5])
