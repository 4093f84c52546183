"""Pins the oracle (parity unpinned by the reference: it has no tests) with
closed forms, exact diagonalisation, autograd and finite differences."""
import itertools

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla
import torch

from oracle import vmc_oracle as vo
from tests.exact_states import ed_ground_state as _ed_ground_state, exact_fc_eigenstate


def _setup(n=8, h=16, L=2, b=12, seed=0):
  rng = np.random.default_rng(seed)
  theta = vo.init_params(n, h, L, rng)
  theta = theta + 0.05 * rng.standard_normal(theta.size).astype(np.float32)  # non-zero biases
  cfg = vo.random_configurations(n, b, np.random.RandomState(seed))
  return theta, cfg


def test_param_count_and_order():
  assert vo.num_params(100, 256, 3) == 157697      # config 3
  assert vo.num_params(36, 128, 3) == 37889        # config 2
  th = np.arange(vo.num_params(3, 2, 2), dtype=np.float32)
  layers = vo.unpack(th, 3, 2, 2)
  assert layers[0][0].shape == (3, 2) and layers[0][0][1, 0] == 2   # row-major w:[in,out]
  assert layers[0][1].tolist() == [6, 7]
  assert layers[2][0].shape == (2, 1) and layers[2][1].shape == (1,)


def test_constant_psi_closed_form():
  cfg = vo.random_configurations(16, 40, np.random.RandomState(3))
  for bonds in (vo.chain_bonds(16), vo.torus_bonds(4, 4)):
    for jx in (1.0, -1.0, 0.3):
      amp = lambda c: np.full(c.shape[0], 3.7, np.float32)
      e = vo.local_value(amp, cfg, bonds, jx, 1.0)
      np.testing.assert_allclose(e, vo.constant_psi_local_energy(cfg, bonds, jx, 1.0), rtol=1e-6)


@pytest.mark.parametrize('n,bonds,jx', [
    (8, vo.chain_bonds(8), 1.0),
    (8, vo.chain_bonds(8), -1.0),
    (12, vo.torus_bonds(4, 3), -1.0),
])
def test_ed_eigenstate_has_constant_local_energy(n, bonds, jx):
  """Role of FullVector (wavefunctions.py:1001-1055): exact state => E_loc == E0."""
  e0, vec, cfgs, index = _ed_ground_state(n, bonds, jx, 1.0)

  def amp(c):
    keys = [tuple(np.nonzero(row < 0)[0]) for row in np.asarray(c)]
    return np.array([vec[index[k]] for k in keys])

  sel = np.abs(vec) > 1e-6
  e = vo.local_value(amp, cfgs[sel], bonds, jx, 1.0, dtype=np.float64)
  np.testing.assert_allclose(e, e0, rtol=0, atol=1e-8)
  if n == 8 and jx == 1.0:
    np.testing.assert_allclose(e0, -3.651093408937176, atol=1e-9)  # 8-site Heisenberg ring


def test_local_value_vs_apply_in_place():
  theta, cfg = _setup()
  amp = lambda c: vo.fc_psi(theta, c, 16, 2, dtype=np.float64)
  bonds = vo.chain_bonds(8)
  lv = vo.local_value(amp, cfg, bonds, -1.0, 1.0, dtype=np.float64)
  ap = vo.apply_in_place(amp, cfg, bonds, -1.0, 1.0, dtype=np.float64)
  np.testing.assert_allclose(ap / amp(cfg), lv, rtol=1e-12)


def _torch_logit(theta_t, x, n, h, L):
  off = 0
  a = x
  for l in range(L + 1):
    fi = n if l == 0 else h
    fo = h if l < L else 1
    w = theta_t[off:off + fi * fo].reshape(fi, fo); off += fi * fo
    b = theta_t[off:off + fo]; off += fo
    a = a @ w + b
    if l < L:
      a = torch.relu(a)
  return a[:, 0]


def test_weighted_grads_vs_autograd_and_fd():
  n, h, L = 8, 16, 2
  theta, cfg = _setup(n, h, L, 12)
  wts = np.random.default_rng(5).standard_normal((12, 2))
  g = vo.weighted_logit_grads(theta, cfg, wts, h, L, dtype=np.float64)
  tt = torch.tensor(theta.astype(np.float64), requires_grad=True)
  x = torch.tensor(cfg.astype(np.float64))
  for c in range(2):
    out = (_torch_logit(tt, x, n, h, L) * torch.tensor(wts[:, c])).sum()
    (ga,) = torch.autograd.grad(out, tt)
    np.testing.assert_allclose(g[c], ga.numpy(), rtol=1e-10, atol=1e-12)
  # central finite differences on a few coordinates
  f = lambda th: (vo.fc_logit(th, cfg, h, L, dtype=np.float64) * wts[:, 0]).sum()
  th64 = theta.astype(np.float64)
  for k in (0, 5, n * h + 3, theta.size - 1, theta.size - 3):
    e = np.zeros_like(th64); e[k] = 1e-6
    fd = (f(th64 + e) - f(th64 - e)) / 2e-6
    np.testing.assert_allclose(g[0][k], fd, rtol=1e-5, atol=1e-7)
  # fp32 twin agrees to fp32 tolerance
  g32 = vo.weighted_logit_grads(theta, cfg, wts, h, L, dtype=np.float32)
  np.testing.assert_allclose(g32, g, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize('act,oact', [('tanh', 'exp'), ('sigmoid', 'tanh'), ('cos', 'sigmoid'),
                                      ('relu', 'identity'), ('identity', 'cos'), ('tan', 'exp')])
def test_general_activation_grads_vs_autograd(act, oact):
  """Every layers.NONLINEARITIES entry as hidden / output activation: the oracle's weighted
  gradients equal torch.autograd of sum_b w_b psi_b / stop_gradient(psi_b) (training.py:545)."""
  n, h, L = 8, 16, 2
  theta, cfg = _setup(n, h, L, 12)
  theta = (0.4 * theta).astype(np.float32); theta[-1] = 0.7
  wts = np.random.default_rng(6).standard_normal((12, 2))
  g = vo.weighted_logit_grads(theta, cfg, wts, h, L, nonlinearity=act, dtype=np.float64,
                              output_activation=oact)
  tf = {'relu': torch.relu, 'exp': torch.exp, 'cos': torch.cos, 'tan': torch.tan,
        'tanh': torch.tanh, 'sigmoid': torch.sigmoid, 'identity': lambda v: v}
  tt = torch.tensor(theta.astype(np.float64), requires_grad=True)
  a = torch.tensor(cfg.astype(np.float64))
  off = 0
  fan = n
  for l in range(L + 1):
    out = h if l < L else 1
    w = tt[off:off + fan * out].reshape(fan, out); off += fan * out
    b = tt[off:off + out]; off += out
    a = a @ w + b
    if l < L:
      a = tf[act](a)
    fan = out
  x = a[:, 0]
  psi = torch.exp(x + 10.0) if oact == 'exp' else tf[oact](x)
  np.testing.assert_allclose(psi.detach().numpy(),
                             vo.fc_psi(theta, cfg, h, L, nonlinearity=act, output_activation=oact,
                                       dtype=np.float64), rtol=1e-12)
  for c in range(2):
    (ga,) = torch.autograd.grad(((psi / psi.detach()) * torch.tensor(wts[:, c])).sum(), tt,
                                retain_graph=True)
    np.testing.assert_allclose(g[c], ga.numpy(), rtol=1e-9, atol=1e-11)


def test_energy_gradient_is_covariance():
  """training.py:560-564 with tf.gradients' batch SUM: grad = B * Cov_b(E, O_k)."""
  n, h, L = 8, 16, 2
  theta, cfg = _setup(n, h, L, 32)
  bonds = vo.chain_bonds(n)
  acc = vo.Accumulators(theta.size, np.float64)
  e = vo.energy_gradient_accumulate(acc, theta, cfg, bonds, -1.0, 1.0, -10.0, h, L, np.float64)
  grad = vo.energy_gradient(acc)
  o = np.stack([vo.weighted_logit_grads(theta, cfg[b:b + 1], np.ones(1), h, L, dtype=np.float64)[0]
                for b in range(32)])
  cov = (e[:, None] * o).mean(0) - e.mean() * o.mean(0)
  np.testing.assert_allclose(grad, 32 * cov, rtol=1e-9, atol=1e-12)


def test_log_overlap_gradient_scale_invariance():
  """grad = mean(G1) - mean(G2)/mean(ratio) must not depend on either shift."""
  n, h, L = 8, 16, 2
  theta, cfg = _setup(n, h, L, 16)
  theta_w = theta + 0.01 * np.random.default_rng(9).standard_normal(theta.size).astype(np.float32)
  bonds = vo.chain_bonds(n)
  out = []
  for s, sw in ((-10.0, -10.0), (-3.0, -10.0), (-10.0, 2.0)):
    acc = vo.Accumulators(theta.size, np.float64)
    vo.log_overlap_accumulate(acc, theta, theta_w, cfg, bonds, -1.0, 1.0, s, sw, 0.12, h, L,
                              np.float64)
    out.append(vo.log_overlap_gradient(acc))
  np.testing.assert_allclose(out[0], out[1], rtol=1e-9, atol=1e-12)
  np.testing.assert_allclose(out[0], out[2], rtol=1e-9, atol=1e-12)


def test_adam_matches_torch_free_formula_and_piecewise_lr():
  assert vo.piecewise_constant(0, [300, 600, 1000], [1e-3, 1e-4, 2e-5, 1e-5]) == 1e-3
  assert vo.piecewise_constant(300, [300, 600, 1000], [1e-3, 1e-4, 2e-5, 1e-5]) == 1e-3
  assert vo.piecewise_constant(301, [300, 600, 1000], [1e-3, 1e-4, 2e-5, 1e-5]) == 1e-4
  assert vo.piecewise_constant(1001, [300, 600, 1000], [1e-3, 1e-4, 2e-5, 1e-5]) == 1e-5
  st = vo.AdamState(3)
  th = np.array([1.0, -2.0, 0.5], np.float32)
  g = np.array([0.1, -0.3, 0.0], np.float32)
  th1 = vo.adam_apply(st, th, g, 1e-3, 0.9, 0.99, 1e-8)
  # first TF1 Adam step: m=(1-b1)g, v=(1-b2)g^2, lr_t=lr*sqrt(1-b2)/(1-b1)
  lr_t = 1e-3 * np.sqrt(1 - 0.99) / (1 - 0.9)
  exp = th - lr_t * (0.1 * g) / (np.sqrt(0.01 * g * g) + 1e-8)
  np.testing.assert_allclose(th1, exp, rtol=1e-5)


def test_update_norm_rule():
  psi = np.array([1.0, 5e12], np.float32)
  s = vo.update_norm(psi, -10.0)
  np.testing.assert_allclose(s, -10 + np.log(5e12) - np.log(1e10), rtol=1e-6)
  assert vo.update_norm(np.array([3.0], np.float32), -10.0) == np.float32(-10.0)


def test_mc_step_conserves_sz_and_accept_rule():
  n, h, L = 12, 16, 2
  theta, cfg = _setup(n, h, L, 64)
  amp = lambda c: vo.fc_psi(theta, c, h, L)
  u, ua = vo.step_uniforms(11, np.arange(64), 0, n)
  i_up, i_dn = vo.propose_exchange(cfg, u)
  new, acc, ratios = vo.mc_step(amp, cfg, i_up, i_dn, ua)
  assert (new.sum(1) == cfg.sum(1)).all()
  assert np.array_equal(acc, ratios > np.sqrt(ua))
  changed = (new != cfg).any(1)
  assert np.array_equal(changed, acc)
  # u = 0 always accepts (ratio > 0), u -> 1 with tiny ratio rejects
  _, acc0, _ = vo.mc_step(amp, cfg, i_up, i_dn, np.zeros(64, np.float32))
  assert acc0.all()


def test_mc_sampling_reproduces_exact_energy():
  """Statistical pin: sampling |psi|^2 of the exact ground state gives E0."""
  n = 8
  bonds = vo.chain_bonds(n)
  e0, vec, cfgs, index = _ed_ground_state(n, bonds, -1.0, 1.0)
  vec = np.abs(vec)

  def amp(c):
    keys = [tuple(np.nonzero(row < 0)[0]) for row in np.asarray(c)]
    return np.array([vec[index[k]] for k in keys])

  cfg = vo.random_configurations(n, 256, np.random.RandomState(1))
  ids = np.arange(256)
  for t in range(40):
    u, ua = vo.step_uniforms(5, ids, t, n)
    i_up, i_dn = vo.propose_exchange(cfg, u)
    cfg, _, _ = vo.mc_step(amp, cfg, i_up, i_dn, ua)
  e = vo.local_value(amp, cfg, bonds, -1.0, 1.0, dtype=np.float64)
  np.testing.assert_allclose(e, e0, atol=1e-8)   # zero-variance for an eigenstate
  # and the sampled distribution is |psi|^2: compare <|S_0^z S_1^z|> with exact
  szsz = (cfg[:, 0] * cfg[:, 1]).mean() / 4
  exact = sum(vec[k] ** 2 * cfgs[k, 0] * cfgs[k, 1] for k in range(len(vec))) / 4
  assert abs(szsz - exact) < 0.03


# ------------------------------------------------------------------ RBM ansatz (wavefunctions.py:391-452)
def test_rbm_logit_is_onsite_plus_sum_log_cosh():
  rng = np.random.default_rng(0)
  for (n, h, L, b) in [(8, 6, 2, 20), (8, 6, 0, 20), (10, 5, 1, 7)]:
    th = vo.rbm_init_params(n, h, L, rng).astype(np.float64)
    th += 0.2 * rng.standard_normal(th.size)
    assert th.size == vo.rbm_num_params(n, h, L) == n + 1 + n * h + h + L * (h * h + h)
    cfg = vo.random_configurations(n, b, np.random.RandomState(1)).astype(np.float64)
    lay = vo.rbm_unpack(th, n, h, L)
    a = cfg
    for (w, bias) in lay[1:-1]:
      a = np.maximum(a @ w + bias, 0)
    z = a @ lay[-1][0] + lay[-1][1]
    ref = (cfg @ lay[0][0] + lay[0][1])[:, 0] + np.log(np.cosh(z)).sum(1)
    np.testing.assert_allclose(vo.rbm_logit(th, cfg, h, L, dtype=np.float64), ref, rtol=1e-13, atol=1e-13)
    # overflow-free where log(cosh(z)) itself overflows
    big = th.copy(); big[-h:] = 800.0
    assert np.isfinite(vo.rbm_logit(big, cfg, h, L, dtype=np.float64)).all()


def test_rbm_weighted_grads_match_finite_differences():
  rng = np.random.default_rng(1)
  for (n, h, L, b) in [(8, 6, 2, 12), (8, 6, 0, 12)]:
    th = vo.rbm_init_params(n, h, L, rng).astype(np.float64)
    th += 0.2 * rng.standard_normal(th.size)
    cfg = vo.random_configurations(n, b, np.random.RandomState(2))
    w = rng.standard_normal((b, 2))
    g = vo.rbm_weighted_logit_grads(th, cfg, w, h, L, dtype=np.float64)
    eps = 1e-6
    for k in range(th.size):
      tp = th.copy(); tp[k] += eps
      tm = th.copy(); tm[k] -= eps
      fd = (vo.rbm_logit(tp, cfg, h, L, dtype=np.float64)
            - vo.rbm_logit(tm, cfg, h, L, dtype=np.float64)) / (2 * eps)
      assert abs(fd @ w[:, 0] - g[0, k]) < 1e-7 and abs(fd @ w[:, 1] - g[1, k]) < 1e-7


def test_rbm_zero_parameters_give_constant_psi_closed_form():
  n, h, L, b = 16, 8, 1, 32
  theta = np.zeros(vo.rbm_num_params(n, h, L), np.float32)
  cfg = vo.random_configurations(n, b, np.random.RandomState(3))
  bonds = vo.torus_bonds(4, 4)
  amp = lambda c: vo.rbm_psi(theta, c, h, L, dtype=np.float64)
  np.testing.assert_allclose(vo.local_value(amp, cfg, bonds, 0.7, 1.0, dtype=np.float64),
                             vo.constant_psi_local_energy(cfg, bonds, 0.7, 1.0), rtol=1e-12)


# --------------------------------------------------------------------------- #
# Convolutional ansatz types (wavefunctions.py:531-615, 710-809; layers.py:89-229)
# --------------------------------------------------------------------------- #
def _torch_periodic_conv(x, w, b):
  """Independent restatement with explicit concat padding (layers.py:118-148) + torch conv2d."""
  import torch
  k = w.shape[0]
  lo = (k - 1) // 2 if k % 2 else k // 2 - 1
  hi = (k - 1) // 2 if k % 2 else k // 2
  xp = torch.cat([x[..., x.shape[3] - lo:], x, x[..., :hi]], 3)
  xp = torch.cat([xp[:, :, xp.shape[2] - lo:], xp, xp[:, :, :hi]], 2)
  return torch.nn.functional.conv2d(xp, w.permute(3, 2, 0, 1), b)


@pytest.mark.parametrize('ansatz,L', [('conv_2d', 3), ('res_net_2d', 2)])
@pytest.mark.parametrize('k', [2, 3, 4, 5])
def test_conv_oracle_against_torch_autograd(ansatz, L, k):
  import torch
  geom = (6, k, 4, 6)
  rng = np.random.default_rng(1)
  th = vo.conv_init_params(ansatz, geom, L, rng).astype(np.float64)
  th += 0.05 * rng.standard_normal(th.size)
  cfg = vo.random_configurations(24, 5, np.random.RandomState(2))
  logit = vo.conv_forward(th, cfg, ansatz, geom, L, 'tanh', np.float64)
  x = torch.tensor(cfg, dtype=torch.float64).reshape(-1, 4, 6, 1).permute(0, 3, 1, 2)
  tl = [(torch.tensor(w, requires_grad=True), torch.tensor(b.copy(), requires_grad=True))
        for w, b in vo.conv_unpack(th, ansatz, geom, L)]
  if ansatz == 'conv_2d':
    a = x
    for l, (w, b) in enumerate(tl):
      a = _torch_periodic_conv(a, w, b)
      if l + 1 != len(tl):
        a = torch.tanh(a)
  else:
    a = _torch_periodic_conv(x, *tl[0])
    for blk in range(L):
      a = a + _torch_periodic_conv(torch.selu(_torch_periodic_conv(a, *tl[1 + 2 * blk])), *tl[2 + 2 * blk])
  t_logit = a.sum((1, 2, 3))
  np.testing.assert_allclose(logit, t_logit.detach().numpy(), rtol=0, atol=1e-12)
  wts = torch.tensor(rng.standard_normal(5))
  (t_logit * wts).sum().backward()
  tg = np.concatenate([np.concatenate([w.grad.numpy().ravel(), b.grad.numpy().ravel()]) for w, b in tl])
  g = vo.ANSATZ[ansatz][2](th, cfg, wts.numpy(), geom, L, nonlinearity='tanh', dtype=np.float64)[0]
  np.testing.assert_allclose(g, tg, rtol=0, atol=1e-11 * max(1.0, np.abs(tg).max()))


def test_conv_even_kernel_padding_is_asymmetric():
  """layers.py:137-141: an even kernel pads k/2 - 1 in front and k/2 behind on both axes (the 1-D
  module does the opposite, layers.py:69-72): a delta kernel at tap (0, 0) of a 2 x 2 kernel
  reads the site itself, tap (1, 1) the site one step up in both axes."""
  x = np.arange(12, dtype=np.float64).reshape(1, 3, 4, 1)
  for tap, shift in (((0, 0), (0, 0)), ((1, 1), (-1, -1)), ((0, 1), (0, -1))):
    w = np.zeros((2, 2, 1, 1)); w[tap] = 1.0
    out = vo.conv2d_periodic(x, w, np.zeros(1))
    np.testing.assert_array_equal(out[0, :, :, 0], np.roll(x[0, :, :, 0], shift, axis=(0, 1)))
  w = np.zeros((3, 3, 1, 1)); w[0, 0] = 1.0       # odd kernel: one site back in both axes
  np.testing.assert_array_equal(vo.conv2d_periodic(x, w, np.zeros(1))[0, :, :, 0],
                                np.roll(x[0, :, :, 0], (1, 1), axis=(0, 1)))


def test_conv_zero_weights_closed_form_and_exchange_symmetry():
  """All-zero kernels: logit = N * sum(b_last) (conv_2d); a translation of the lattice leaves the
  amplitude unchanged (periodic convolutions + global sum)."""
  geom, L = (4, 3, 4, 4), 2
  th = np.zeros(vo.conv_num_params('conv_2d', geom, L))
  th[-4:] = [0.1, 0.2, 0.3, 0.4]
  cfg = vo.random_configurations(16, 3, np.random.RandomState(0))
  np.testing.assert_allclose(vo.conv_forward(th, cfg, 'conv_2d', geom, L, dtype=np.float64), 16 * 1.0)
  for ansatz in ('conv_2d', 'res_net_2d'):
    th = vo.conv_init_params(ansatz, geom, L, np.random.default_rng(4)).astype(np.float64)
    a = vo.conv_forward(th, cfg, ansatz, geom, L, dtype=np.float64)
    shifted = np.roll(cfg.reshape(-1, 4, 4), (1, 2), axis=(1, 2)).reshape(-1, 16)
    np.testing.assert_allclose(vo.conv_forward(th, shifted, ansatz, geom, L, dtype=np.float64), a, atol=1e-12)


@pytest.mark.parametrize('ansatz,L', [('conv_1d', 3), ('res_net_1d', 2)])
@pytest.mark.parametrize('k', [2, 3, 4, 5])
def test_conv1d_oracle_against_torch_autograd(ansatz, L, k):
  """Conv1DNetwork / ResNet1D (wavefunctions.py:455-527, 618-707) against explicit concat padding
  (layers.py:51-74: an even kernel pads k/2 in front and k/2 - 1 behind) + torch conv1d + autograd."""
  import torch
  n, geom = 12, (6, k, 12, 1)
  rng = np.random.default_rng(1)
  th = vo.conv_init_params(ansatz, geom, L, rng).astype(np.float64)
  th += 0.05 * rng.standard_normal(th.size)
  cfg = vo.random_configurations(n, 5, np.random.RandomState(2))
  logit = vo.conv_forward(th, cfg, ansatz, geom, L, 'tanh', np.float64)
  x = torch.tensor(cfg, dtype=torch.float64).reshape(-1, 1, n)
  tl = [(torch.tensor(w[:, 0], requires_grad=True), torch.tensor(b.copy(), requires_grad=True))
        for w, b in vo.conv_unpack(th, ansatz, geom, L)]

  def pconv(x, w, b):
    kk = w.shape[0]
    lo = (kk - 1) // 2 if kk % 2 else kk // 2
    hi = (kk - 1) // 2 if kk % 2 else kk // 2 - 1
    xp = torch.cat([x[..., x.shape[2] - lo:], x, x[..., :hi]], 2)
    return torch.nn.functional.conv1d(xp, w.permute(2, 1, 0), b)

  if ansatz == 'conv_1d':
    a = x
    for l, (w, b) in enumerate(tl):
      a = pconv(a, w, b)
      if l + 1 != len(tl):
        a = torch.tanh(a)
  else:
    a = pconv(x, *tl[0])
    for blk in range(L):
      a = a + pconv(torch.selu(pconv(a, *tl[1 + 2 * blk])), *tl[2 + 2 * blk])
  t_logit = a.sum((1, 2))
  np.testing.assert_allclose(logit, t_logit.detach().numpy(), rtol=0, atol=1e-12)
  wts = torch.tensor(rng.standard_normal(5))
  (t_logit * wts).sum().backward()
  tg = np.concatenate([np.concatenate([w.grad.numpy().ravel(), b.grad.numpy().ravel()]) for w, b in tl])
  g = vo.ANSATZ[ansatz][2](th, cfg, wts.numpy(), geom, L, nonlinearity='tanh', dtype=np.float64)[0]
  np.testing.assert_allclose(g, tg, rtol=0, atol=1e-11 * max(1.0, np.abs(tg).max()))


@pytest.mark.parametrize('n,h,L', [(8, 128, 2), (10, 256, 3)])
def test_exact_fc_eigenstate_has_constant_local_energy_and_zero_gradient(n, h, L):
  """A FullyConnectedNetwork that equals the ED ground state (tests/exact_states.py): E_loc == E0 on
  the whole sector, and the energy gradient <E O> - <E><O> (training.py:560-564) vanishes."""
  bonds = vo.chain_bonds(n)
  theta, e0, cfgs, vec = exact_fc_eigenstate(n, bonds, h, L)
  amp64 = lambda c: vo.fc_psi(theta.astype(np.float64), c, h, L, shift=0.0, dtype=np.float64)
  e64 = vo.local_value(amp64, cfgs.astype(np.float64), bonds, -1.0, 1.0, dtype=np.float64)
  assert np.abs(e64 - e0).max() < 1e-4          # fp32-rounded parameters, fp64 arithmetic
  amp32 = lambda c: vo.fc_psi(theta, c, h, L, shift=0.0)
  e32 = vo.local_value(amp32, cfgs, bonds, -1.0, 1.0)
  assert np.abs(e32 - e0).max() < 2e-3          # fp32 arithmetic
  # zero-variance principle: with E_loc constant the covariance gradient is zero whatever the sampling
  acc = vo.Accumulators(theta.size, np.float64)
  vo.energy_gradient_accumulate(acc, theta, cfgs, bonds, -1.0, 1.0, 0.0, h, L, dtype=np.float64)
  g = vo.energy_gradient(acc)
  scale = np.abs(acc.g2_total / acc.g_count).max()
  assert np.abs(g).max() < 1e-4 * scale


def test_tf1_adam_is_torch_adam_with_a_rescaled_epsilon():
  """An independent implementation of the optimizer the oracle restates (training.py:76-91,
  tf.train.AdamOptimizer): TF1 applies  theta -= lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps),
  torch.optim.Adam  theta -= lr / (1 - b1^t) * m / (sqrt(v / (1 - b2^t)) + eps').  They are the same
  update when eps' = eps / sqrt(1 - b2^t); with that epsilon set per step torch's optimizer must
  reproduce vo.adam_apply over several steps of changing gradients."""
  import torch
  rng = np.random.default_rng(3)
  theta0 = rng.standard_normal(50).astype(np.float32)
  lr, b1, b2, eps = 1e-3, 0.9, 0.99, 1e-8
  p = torch.nn.Parameter(torch.tensor(theta0.astype(np.float64)))
  opt = torch.optim.Adam([p], lr=lr, betas=(b1, b2), eps=eps)
  st = vo.AdamState(theta0.size)
  theta = theta0.copy()
  for t in range(1, 8):
    g = (rng.standard_normal(50) * (10.0 ** rng.integers(-3, 2))).astype(np.float32)
    theta = vo.adam_apply(st, theta, g, lr, b1, b2, eps)
    opt.param_groups[0]['eps'] = eps / np.sqrt(1.0 - b2 ** t)
    p.grad = torch.tensor(g.astype(np.float64))
    opt.step()
    np.testing.assert_allclose(theta, p.detach().numpy(), rtol=0, atol=2e-6)
