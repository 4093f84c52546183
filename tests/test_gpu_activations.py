"""GPU parity of every entry of layers.NONLINEARITIES (layers.py:13-21) as hidden activation and
as output activation (wavefunctions.py:350-353: any output activation other than exp means
psi = g(x) with no exp_norm_shift, and every ratio is taken in the linear domain).

Tolerances (stated): as tests/test_gpu_engine.py for relu / tanh / sigmoid / identity / exp; the
cosine and tangent go through the hardware v_cos_f32 / v_sin_f32 (|error| ~ 1e-6 per unit), so
their logits are compared at 2e-4 and their energies / gradients at 2e-3 / 1e-2.
"""
import numpy as np
import pytest

from oracle import vmc_oracle as vo

pytestmark = pytest.mark.gpu

HIDDEN = ['tanh', 'sigmoid', 'identity', 'exp', 'cos', 'tan']
SHAPES = [
    (16, 32, 2, 64, 'torus4x4'),
    (36, 128, 3, 100, 'torus6x6'),
    (12, 200, 1, 48, 'chain'),      # single layer, H padded to 256: k_tail0
    (20, 256, 3, 40, 'chain'),      # 8-wave sampler
    (16, 320, 2, 40, 'torus4x4'),   # 257 .. 512 units: k_sweep16<24>, k_tail_lds<24>, k_backprop16<24>
    (12, 448, 3, 30, 'chain'),      # ... <32>, two H x H layers
]


def _bonds(kind, n):
  if kind == 'chain':
    return vo.chain_bonds(n)
  lx = int(kind[5])
  return vo.torus_bonds(lx, n // lx)


def _tols(act, h=0):
  # exp as HIDDEN activation beyond 256 units: three layers of up to 512 positive terms e^z each,
  # summed in fp32 in an order that differs from the oracle's -- the loose class as well
  loose = act in ('cos', 'tan') or (act == 'exp' and h > 256)
  return (2e-4 if loose else 2e-5), (2e-3 if loose else 2e-4), (1e-2 if loose else 2e-3)


def _close(a, b, rel, floor=1.0):
  a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
  tol = rel * np.maximum(floor, np.abs(b))
  assert (np.abs(a - b) <= tol).all(), 'max err {} (tol {})'.format(
      np.abs(a - b).max(), tol[np.argmax(np.abs(a - b) - tol)])


def _make(n, h, L, b, kind, act, oact='exp', seed=0, scale=1.0, b_out=0.0):
  from cgs_vmc_amd.engine import VmcEngine
  rng = np.random.default_rng(seed)
  theta = (scale * vo.init_params(n, h, L, rng)).astype(np.float32)
  theta += (0.03 * rng.standard_normal(theta.size)).astype(np.float32)
  theta[-1] = b_out
  cfg = vo.random_configurations(n, b, np.random.RandomState(seed + 1))
  bonds = _bonds(kind, n)
  eng = VmcEngine(n, b, L, h, nonlinearity=act, output_activation=oact, seed=2024)
  eng.set_params(theta)
  eng.set_configs(cfg)
  eng.set_bonds(bonds, -1.0, 1.0)
  return eng, theta, cfg, bonds


@pytest.mark.parametrize('n,h,L,b,kind', SHAPES)
@pytest.mark.parametrize('act', HIDDEN)
def test_hidden_activation_parity(act, n, h, L, b, kind):
  from cgs_vmc_amd import _hip
  t_logit, t_e, t_g = _tols(act, h)
  scale = {'tan': 0.2, 'exp': 0.4}.get(act, 1.0)     # keep tan away from its poles, exp from overflow
  eng, theta, cfg, bonds = _make(n, h, L, b, kind, act, scale=scale)
  assert eng.kernel_path() == (0 if h <= 256 else 1)
  kw = dict(nonlinearity=act)
  amp = lambda c: vo.fc_psi(theta, c, h, L, dtype=np.float64, **kw)
  _close(eng.amplitude()[0], vo.fc_logit(theta, cfg, h, L, dtype=np.float64, **kw), t_logit)
  _close(eng.amplitude(cfg[:7])[0], vo.fc_logit(theta, cfg[:7], h, L, dtype=np.float64, **kw), t_logit)
  _close(eng.local_energy()[0], vo.local_value(amp, cfg, bonds, -1.0, 1.0, dtype=np.float64), t_e)
  # gradient accumulators (before the chains move: the sampler hands nothing over yet)
  eng.reset_accumulators()
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  acc = vo.Accumulators(theta.size, np.float64)
  vo.energy_gradient_accumulate(acc, theta, cfg, bonds, -1.0, 1.0, -10.0, h, L, np.float64, **kw)
  got = eng.get_accumulators().astype(np.float64)
  p = theta.size
  for lo, ref in ((0, acc.g1_total), (p, acc.g2_total)):
    assert np.abs(got[lo:lo + p] - ref).max() <= t_g * np.abs(ref).max() + 1e-4
  # injected mc_steps: accept masks, moved chains, cache
  cur = cfg
  for step in range(3):
    u_sites, u_acc = vo.step_uniforms(99, np.arange(b), step, n)
    i_up, i_dn = vo.propose_exchange(cur, u_sites)
    _, acc_ref, ratios = vo.mc_step(amp, cur, i_up, i_dn, u_acc)
    mask = eng.mc_step_injected(i_up, i_dn, u_acc)
    band = np.abs(ratios - np.sqrt(u_acc.astype(np.float64))) < 10 * t_logit * np.maximum(ratios, 1e-30)
    assert np.array_equal(mask[~band], acc_ref[~band])
    expect = cur.copy()
    rows = np.arange(b)[mask]
    expect[rows, i_dn[mask]] = 1.0
    expect[rows, i_up[mask]] = -1.0
    np.testing.assert_array_equal(eng.get_configs(), expect)
    cur = expect
    _close(eng.amplitude()[0], vo.fc_logit(theta, cur, h, L, dtype=np.float64, **kw), t_logit)
  # the sampler's own stream: Sz conserved, exact cache, activations handed to the gradient path
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)     # a gradient accumulate was seen: hand-over on
  eng.mc_steps(2 * n)
  moved = eng.get_configs()
  assert (moved.sum(1) == cfg.sum(1)).all() and (np.abs(moved) == 1).all()
  _close(eng.amplitude()[0], vo.fc_logit(theta, moved, h, L, dtype=np.float64, **kw), t_logit)
  eng.reset_accumulators()
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  acc = vo.Accumulators(theta.size, np.float64)
  vo.energy_gradient_accumulate(acc, theta, moved, bonds, -1.0, 1.0, -10.0, h, L, np.float64, **kw)
  g = eng.get_gradient(_hip.VMC_MODE_ENERGY_GRADIENT)
  gref = vo.energy_gradient(acc)
  assert np.abs(g - gref).max() <= t_g * np.abs(gref).max() + 2e-4
  eng.close()


@pytest.mark.parametrize('act', ['relu', 'tanh'])
@pytest.mark.parametrize('oact', ['tanh', 'sigmoid', 'identity', 'relu', 'cos', 'tan'])
def test_output_activation_parity(oact, act):
  """psi = g(x): amplitudes, signed local-energy ratios, |psi'|/|psi| accept rule, O_k with the
  g'(x)/g(x) factor, update_norm a no-op, the log-overlap accumulators."""
  from cgs_vmc_amd import _hip
  n, h, L, b, kind = 16, 64, 2, 96, 'torus4x4'
  t_logit, t_e, t_g = _tols(oact if oact in ('cos', 'tan') else act)
  # x is kept away from the zeros of g (b_out) so that 1/psi stays tame
  eng, theta, cfg, bonds = _make(n, h, L, b, kind, act, oact, scale=0.5, b_out=0.8)
  kw = dict(nonlinearity=act, output_activation=oact)
  amp = lambda c: vo.fc_psi(theta, c, h, L, dtype=np.float64, **kw)
  logit, psi = eng.amplitude()
  _close(logit, vo.fc_logit(theta, cfg, h, L, nonlinearity=act, dtype=np.float64), t_logit)
  _close(psi, amp(cfg), 10 * t_logit)
  _close(eng.local_energy()[0], vo.local_value(amp, cfg, bonds, -1.0, 1.0, dtype=np.float64), t_e)
  shift = eng.get_shift()
  eng.update_norm(1e-3)                     # wavefunctions.py:276-277: None without an exp output
  assert eng.get_shift() == shift
  np.testing.assert_array_equal(eng.amplitude()[1], psi)
  eng.reset_accumulators()
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  acc = vo.Accumulators(theta.size, np.float64)
  vo.energy_gradient_accumulate(acc, theta, cfg, bonds, -1.0, 1.0, -10.0, h, L, np.float64, **kw)
  got = eng.get_accumulators().astype(np.float64)
  p = theta.size
  for lo, ref in ((0, acc.g1_total), (p, acc.g2_total)):
    assert np.abs(got[lo:lo + p] - ref).max() <= t_g * np.abs(ref).max() + 1e-4
  # log-overlap accumulators against a perturbed supervisor
  theta_w = (theta + 0.02 * np.random.default_rng(5).standard_normal(theta.size)).astype(np.float32)
  theta_w[-1] = theta[-1]
  eng.set_params(theta_w, _hip.VMC_OMEGA)
  eng.reset_accumulators()
  eng.accumulate(_hip.VMC_MODE_LOG_OVERLAP_ITSWO, 0.12)
  acc = vo.Accumulators(theta.size, np.float64)
  vo.log_overlap_accumulate(acc, theta, theta_w, cfg, bonds, -1.0, 1.0, -10.0, -10.0, 0.12, h, L,
                            np.float64, **kw)
  g = eng.get_gradient(_hip.VMC_MODE_LOG_OVERLAP_ITSWO)
  gref = vo.log_overlap_gradient(acc)
  assert np.abs(g - gref).max() <= 5 * t_g * np.abs(gref).max() + 2e-4
  # accept rule in the linear domain
  cur = cfg
  for step in range(3):
    u_sites, u_acc = vo.step_uniforms(7, np.arange(b), step, n)
    i_up, i_dn = vo.propose_exchange(cur, u_sites)
    _, acc_ref, ratios = vo.mc_step(amp, cur, i_up, i_dn, u_acc)
    mask = eng.mc_step_injected(i_up, i_dn, u_acc)
    band = np.abs(ratios - np.sqrt(u_acc.astype(np.float64))) < 20 * t_logit * np.maximum(ratios, 1e-30)
    assert np.array_equal(mask[~band], acc_ref[~band])
    cur = eng.get_configs()
  eng.mc_steps(n)
  moved = eng.get_configs()
  assert (moved.sum(1) == cfg.sum(1)).all()
  _close(eng.amplitude()[1], amp(moved), 10 * t_logit)
  eng.close()


def test_activation_error_behaviour():
  from cgs_vmc_amd.engine import VmcEngine
  with pytest.raises(ValueError):
    VmcEngine(16, 8, 2, 32, nonlinearity='gelu')
  with pytest.raises(ValueError):
    VmcEngine(16, 8, 2, 32, output_activation='tanh', ansatz='rbm')
  eng = VmcEngine(16, 8, 2, 32, output_activation='tanh')
  with pytest.raises(NotImplementedError):
    eng.sr_reserve(2)                       # SR (an extension) needs the exp output
  eng.close()
