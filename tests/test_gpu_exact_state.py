"""GPU known-answer test that does not go through the oracle's restatement of the network: a
fully_connected ansatz whose parameters are solved (tests/exact_states.py, fp64 ED + least squares)
so that psi IS the Heisenberg ground state on the whole Sz = 0 sector.  Then, on the HIP path:
E_loc(R) == E0 for every configuration before and after sampling, the covariance gradient vanishes,
and the sampler's stationary distribution reproduces the exact nearest-neighbour correlation."""
import numpy as np
import pytest

from oracle import vmc_oracle as vo
from tests.exact_states import exact_fc_eigenstate

pytestmark = pytest.mark.gpu


# (8 sites, 2 x 128): one H x H layer on the 4-wave kernels; (10 sites, 3 x 256): the config-3 kernels;
# round 4: the 4 x 2 torus (a 2-D bond list with doubled bonds across the short direction), the fused
# 384- and 512-unit kernels (k_sweep16<24|32>, k_tail_lds, k_backprop16<24|32>), and the config-3 shape with
# the local-energy rows on the 3 x bf16 split kernel (CGS_VMC_SPLIT_BF16=1)
CASES = [(8, 128, 2, 'chain', False), (10, 256, 3, 'chain', False), (8, 128, 2, 'torus4x2', False),
         (10, 384, 2, 'chain', False), (10, 512, 3, 'chain', False), (10, 256, 3, 'chain', True),
         (10, 256, 3, 'chain', 2),           # round 5: row kernel AND sampler on the split (k_tail16r + k_sweep16s)
         (10, 640, 2, 'chain', False)]       # beyond 512 units: the general path (128 x 128-tile GEMM, k_wide_accept)


@pytest.mark.parametrize('n,h,L,kind,split', CASES)
def test_exact_eigenstate_on_the_hip_path(monkeypatch, n, h, L, kind, split):
  from cgs_vmc_amd import _hip
  from cgs_vmc_amd.engine import VmcEngine
  if split:
    monkeypatch.setenv('CGS_VMC_SPLIT_BF16', str(int(split)))
  else:
    monkeypatch.delenv('CGS_VMC_SPLIT_BF16', raising=False)
  bonds = vo.chain_bonds(n) if kind == 'chain' else vo.torus_bonds(4, 2)
  theta, e0, cfgs, vec = exact_fc_eigenstate(n, bonds, h, L)
  b = len(cfgs)                                   # 70 / 252: every configuration of the sector once
  eng = VmcEngine(n, b, L, h, seed=5)
  assert eng.kernel_path() == ({1: 4, 2: 5}[int(split)] if split else (2 if h > 512 else (1 if h > 256 else 0)))
  eng.set_params(theta)
  eng.set_shift(0.0)
  eng.set_configs(cfgs)
  eng.set_bonds(bonds, -1.0, 1.0)

  logit = eng.amplitude()[0]
  assert np.abs(logit - np.log(vec)).max() < 2e-4
  e = eng.local_energy()[0]
  assert np.abs(e - e0).max() < 2e-3              # fp32 evaluation of an fp64-exact eigenstate

  # zero-variance principle (training.py:560-564): <E O> - <E><O> = 0 when E_loc is constant
  eng.reset_accumulators()
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  g = eng.get_gradient(_hip.VMC_MODE_ENERGY_GRADIENT)
  acc = eng.get_accumulators()
  p = theta.size
  scale = np.abs(acc[p:2 * p]).max() / b          # |mean of E_loc O_k|, the size of either term
  assert np.abs(g).max() < 2e-3 * scale
  assert abs(eng.mean_energy() - e0) < 2e-3

  # imaginary-time target of an eigenstate is the eigenstate: with omega = psi the ratio
  # (1 - beta H) omega / psi = 1 - beta E0 is constant and the log-overlap gradient
  # (training.py:652-729) vanishes for any set of samples
  eng.transfer_params()
  eng.set_shift(0.0, _hip.VMC_OMEGA)
  eng.reset_accumulators()
  eng.accumulate(_hip.VMC_MODE_LOG_OVERLAP_ITSWO, 0.12)
  g = eng.get_gradient(_hip.VMC_MODE_LOG_OVERLAP_ITSWO)
  acc = eng.get_accumulators()
  scale = np.abs(acc[:p]).max() / b               # |mean of O_k|
  assert np.abs(g).max() < 2e-3 * scale

  # sampling: E_loc stays E0 on every chain, and the chains sample |psi|^2: the exact
  # nearest-neighbour correlation <s_i s_i+1> = sum_R psi(R)^2 s_i s_i+1 (translation invariant)
  exact = float(np.mean([(vec ** 2 * cfgs[:, i] * cfgs[:, j]).sum() for (i, j) in bonds]))
  eng.mc_steps(20 * n)
  snaps = 150
  corr = 0.0
  for _ in range(snaps):
    eng.mc_steps(n)
    c = eng.get_configs()
    corr += np.mean([np.mean(c[:, i] * c[:, j]) for (i, j) in bonds])
  corr /= snaps
  # b * snaps * n bond samples, strongly correlated within a configuration: ~ b * snaps independent
  sigma = 1.0 / np.sqrt(b * snaps)
  assert abs(corr - exact) < 5 * sigma, (corr, exact)
  e = eng.local_energy()[0]
  assert np.abs(e - e0).max() < 2e-3
  out = eng.get_configs()
  assert (out.sum(1) == 0).all()
  eng.close()


# the convolutional path against the same kind of known answer (round 4): parameters solved so that the
# network IS the ground state (tests/exact_states.py:exact_conv_eigenstate -- the last convolution is linear in
# its weights and the logit only sees the site sums of its input); one to four channel blocks, 3 .. 9 taps
CONV_CASES = [('conv_2d', (16, 3, 4, 2), 3, 'torus'), ('conv_1d', (48, 5, 8, 1), 3, 'chain'),
              ('conv_2d', (64, 3, 4, 2), 3, 'torus'), ('conv_1d', (24, 9, 8, 1), 3, 'chain'),
              # ResNet2D / ResNet1D (one and two blocks): the shortcut's site sum moves to the right-hand side
              ('res_net_2d', (16, 3, 4, 2), 1, 'torus'), ('res_net_1d', (32, 5, 8, 1), 2, 'chain')]


@pytest.mark.parametrize('ansatz,geom,L,kind', CONV_CASES)
def test_exact_eigenstate_on_the_convolutional_path(ansatz, geom, L, kind):
  from cgs_vmc_amd import _hip
  from cgs_vmc_amd.engine import VmcEngine
  from tests.exact_states import exact_conv_eigenstate
  f, k, sx, sy = geom
  n = sx * sy
  bonds = vo.chain_bonds(n) if kind == 'chain' else vo.torus_bonds(sy, sx)
  theta, e0, cfgs, vec = exact_conv_eigenstate(ansatz, geom, L, bonds)
  b = len(cfgs)
  eng = VmcEngine(n, b, L, f, seed=5, ansatz=ansatz, kernel_size=k, size_x=sx, size_y=sy)
  eng.set_params(theta)
  eng.set_shift(0.0)
  eng.set_configs(cfgs)
  eng.set_bonds(bonds, -1.0, 1.0)
  # (measured on MI355X: logits within 3e-6 of log psi_ED, local energies within 3e-5 of E0, the gradient
  # within 1e-5 of the size of its two terms -- fp32 evaluation of an fp64-exact eigenstate)
  assert np.abs(eng.amplitude()[0] - np.log(vec)).max() < 2e-5
  e = eng.local_energy()[0]
  assert np.abs(e - e0).max() < 2e-4
  # zero-variance principle (training.py:560-564): <E O> - <E><O> = 0 when E_loc is constant
  eng.reset_accumulators()
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  g = eng.get_gradient(_hip.VMC_MODE_ENERGY_GRADIENT)
  acc = eng.get_accumulators()
  p = theta.size
  size = np.abs(acc[p:2 * p]).max() / b               # |mean of E_loc O_k|, the size of either term
  assert np.abs(g).max() < 1e-4 * size
  assert abs(eng.mean_energy() - e0) < 2e-4
  # imaginary-time target of an eigenstate is the eigenstate (training.py:652-729): with omega = psi the ratio
  # (1 - beta H) omega / psi is constant and the log-overlap gradient vanishes for any set of samples
  eng.transfer_params()
  eng.set_shift(0.0, _hip.VMC_OMEGA)
  eng.reset_accumulators()
  eng.accumulate(_hip.VMC_MODE_LOG_OVERLAP_ITSWO, 0.12)
  g = eng.get_gradient(_hip.VMC_MODE_LOG_OVERLAP_ITSWO)
  acc = eng.get_accumulators()
  assert np.abs(g).max() < 1e-4 * (np.abs(acc[:p]).max() / b)
  # sampling keeps E_loc = E0 on every chain and the chains inside the sector, and the chains sample |psi|^2:
  # the exact nearest-neighbour correlation sum_R psi(R)^2 s_i s_j, averaged over the bonds
  exact = float(np.mean([(vec ** 2 * cfgs[:, i] * cfgs[:, j]).sum() for (i, j) in bonds]))
  eng.mc_steps(20 * n)
  snaps, corr = 150, 0.0
  for _ in range(snaps):
    eng.mc_steps(n)
    c = eng.get_configs()
    corr += np.mean([np.mean(c[:, i] * c[:, j]) for (i, j) in bonds])
  corr /= snaps
  sigma = 1.0 / np.sqrt(b * snaps)        # strongly correlated within a configuration: ~ b * snaps independent samples
  assert abs(corr - exact) < 5 * sigma, (corr, exact)
  e = eng.local_energy()[0]
  assert np.abs(e - e0).max() < 2e-4
  assert (eng.get_configs().sum(1) == 0).all()
  eng.close()
