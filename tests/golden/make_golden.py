"""Generates tests/golden/vmc_small.npz from the numpy oracle (oracle/vmc_oracle.py).

PROVENANCE: the reference (TensorFlow 1.x + Sonnet) cannot be imported in the build
container, so NO vector in this file originates from the reference's code; they are the
oracle's float64 outputs on fixed seeded inputs.  The oracle itself is pinned
by tests/test_oracle_physics.py (closed forms, exact diagonalisation, autograd, finite
differences) and tests/test_oracle_rng.py (Random123 known answers).

  python tests/golden/make_golden.py        # rewrites every fixture (--wide-only: the round-3 ones)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import vmc_oracle as vo  # noqa: E402

CASES = {
    # name: (n_sites, H, L, B, bonds)
    'chain16': (16, 32, 2, 64, vo.chain_bonds(16)),          # BASELINE config 1 (default chain)
    'torus4x4': (16, 32, 2, 64, vo.torus_bonds(4, 4)),       # BASELINE config 1 (J.txt torus)
    'torus6x6': (36, 64, 3, 48, vo.torus_bonds(6, 6)),       # BASELINE config 2 lattice, 3 layers
}
SEED, JX, JZ, BETA = 2024, -1.0, 1.0, 0.12


def build_case(name):
  n, h, L, b, bonds = CASES[name]
  rng = np.random.default_rng(abs(hash(name)) % 1000 + 11) if False else np.random.default_rng(
      {'chain16': 11, 'torus4x4': 12, 'torus6x6': 13}[name])
  theta = vo.init_params(n, h, L, rng)
  theta = (theta + 0.05 * rng.standard_normal(theta.size)).astype(np.float32)
  theta_w = (theta + 0.02 * rng.standard_normal(theta.size)).astype(np.float32)
  cfg = vo.random_configurations(n, b, np.random.RandomState(7))
  f64 = np.float64
  amp = lambda c: vo.fc_psi(theta, c, h, L, dtype=f64)
  out = dict(theta=theta, theta_omega=theta_w, configs=cfg,
             bonds=np.asarray(bonds, np.int32), shape=np.array([n, h, L, b]),
             seed=np.array([SEED]), couplings=np.array([JX, JZ, BETA]))
  out['logit'] = vo.fc_logit(theta, cfg, h, L, dtype=f64)
  diag, off = vo.heisenberg_build(amp, cfg, bonds, JX, JZ, f64)
  out['diag'], out['offdiag_over_psi'] = diag, off / amp(cfg)
  out['eloc'] = diag + off / amp(cfg)
  # proposals + one injected step per listed absolute step
  steps = np.array([0, 1, 5, 123456789012], np.uint64)
  ups, dns, us, accs, ratios = [], [], [], [], []
  for s in steps:
    u_sites, u_acc = vo.step_uniforms(SEED, np.arange(b), int(s), n)
    i_up, i_dn = vo.propose_exchange(cfg, u_sites)
    _, acc, ratio = vo.mc_step(amp, cfg, i_up, i_dn, u_acc)
    ups.append(i_up); dns.append(i_dn); us.append(u_acc); accs.append(acc); ratios.append(ratio)
  out.update(steps=steps, i_up=np.array(ups, np.int32), i_dn=np.array(dns, np.int32),
             u_accept=np.array(us, np.float32), accept=np.array(accs), ratio=np.array(ratios))
  # 3 sweeps of the sampler from the oracle (chains after, for the statistically exact part
  # of the trajectory see tests)
  acc_eg = vo.Accumulators(theta.size, f64)
  vo.energy_gradient_accumulate(acc_eg, theta, cfg, bonds, JX, JZ, -10.0, h, L, f64)
  out['eg_g1'], out['eg_g2'] = acc_eg.g1_total, acc_eg.g2_total
  out['eg_scalars'] = np.array([acc_eg.e_total, acc_eg.e_count, acc_eg.g_count])
  out['eg_grad'] = vo.energy_gradient(acc_eg)
  st = vo.AdamState(theta.size)
  out['eg_theta_after_adam'] = vo.adam_apply(st, theta, out['eg_grad'], 1e-3, 0.9, 0.99, 1e-8)
  acc_it = vo.Accumulators(theta.size, f64)
  vo.log_overlap_accumulate(acc_it, theta, theta_w, cfg, bonds, JX, JZ, -9.0, -10.0, BETA, h, L, f64)
  out['it_g1'], out['it_g2'] = acc_it.g1_total, acc_it.g2_total
  out['it_scalars'] = np.array([acc_it.e_total, acc_it.e_count, acc_it.r_total, acc_it.r_count])
  out['it_grad'] = vo.log_overlap_gradient(acc_it)
  return out


RBM_CASES = {
    # name: (n_sites, H, num_layers, B, bonds) of RestrictedBoltzmannNetwork
    'rbm_torus4x4': (16, 32, 1, 64, vo.torus_bonds(4, 4)),
    'rbm_classic_chain12': (12, 24, 0, 40, vo.chain_bonds(12)),   # num_layers = 0
}


# more than 256 hidden units (round 3: the fused 257..512 path; tests/golden/rbm_wide.npz)
RBM_WIDE_CASES = {
    'rbm_classic_chain12_h400': (12, 400, 0, 24, vo.chain_bonds(12)),   # alpha = 33: k_sweep16<32> + k_tail0
    'rbm_chain10_h260_l1': (10, 260, 1, 20, vo.chain_bonds(10)),        # padded to 384: k_tail_lds<24, RBM>
}
_RBM_SEEDS = {'rbm_torus4x4': 21, 'rbm_classic_chain12': 22, 'rbm_classic_chain12_h400': 23,
              'rbm_chain10_h260_l1': 24}


def build_rbm_case(name, cases=None):
  """Same quantities as build_case for the rbm ansatz (tests/golden/rbm_small.npz, rbm_wide.npz)."""
  n, h, L, b, bonds = (cases or RBM_CASES)[name]
  rng = np.random.default_rng(_RBM_SEEDS[name])
  theta = vo.rbm_init_params(n, h, L, rng)
  theta = (theta + 0.05 * rng.standard_normal(theta.size)).astype(np.float32)
  cfg = vo.random_configurations(n, b, np.random.RandomState(7))
  f64 = np.float64
  amp = lambda c: vo.rbm_psi(theta, c, h, L, dtype=f64)
  out = dict(theta=theta, configs=cfg, bonds=np.asarray(bonds, np.int32),
             shape=np.array([n, h, L, b]), seed=np.array([SEED]), couplings=np.array([JX, JZ, BETA]))
  out['logit'] = vo.rbm_logit(theta, cfg, h, L, dtype=f64)
  out['eloc'] = vo.local_value(amp, cfg, bonds, JX, JZ, dtype=f64)
  u_sites, u_acc = vo.step_uniforms(SEED, np.arange(b), 3, n)
  i_up, i_dn = vo.propose_exchange(cfg, u_sites)
  _, acc, ratio = vo.mc_step(amp, cfg, i_up, i_dn, u_acc)
  out.update(i_up=i_up.astype(np.int32), i_dn=i_dn.astype(np.int32), u_accept=u_acc, accept=acc,
             ratio=ratio)
  acc_eg = vo.Accumulators(theta.size, f64)
  vo.energy_gradient_accumulate(acc_eg, theta, cfg, bonds, JX, JZ, -10.0, h, L, f64, ansatz='rbm')
  if cases is None:                    # the wide fixtures keep the gradient only (file size)
    out['eg_g1'], out['eg_g2'] = acc_eg.g1_total, acc_eg.g2_total
  out['eg_grad'] = vo.energy_gradient(acc_eg)
  return out


CONV_CASES = {
    # name: (ansatz, geom = (filters, kernel, size_x, size_y), num_layers / blocks, B, bonds, nonlinearity)
    'conv2d_4x4': ('conv_2d', (8, 3, 4, 4), 2, 32, vo.torus_bonds(4, 4), 'relu'),
    'conv2d_6x4_even': ('conv_2d', (16, 4, 6, 4), 3, 24, vo.torus_bonds(4, 6), 'tanh'),
    'resnet2d_4x4': ('res_net_2d', (8, 3, 4, 4), 1, 32, vo.torus_bonds(4, 4), 'relu'),
    'conv1d_12_even': ('conv_1d', (8, 4, 12, 1), 3, 24, vo.chain_bonds(12), 'relu'),
    'resnet1d_12': ('res_net_1d', (16, 5, 12, 1), 2, 24, vo.chain_bonds(12), 'relu'),
}


# more than 16 filters and the cosine (round 3; tests/golden/conv_wide.npz)
CONV_WIDE_CASES = {
    'conv2d_4x4_f32': ('conv_2d', (32, 3, 4, 4), 3, 24, vo.torus_bonds(4, 4), 'relu'),
    'conv2d_6x4_f24_cos': ('conv_2d', (24, 3, 6, 4), 2, 20, vo.torus_bonds(4, 6), 'cos'),
    'resnet2d_4x4_f32': ('res_net_2d', (32, 3, 4, 4), 1, 24, vo.torus_bonds(4, 4), 'relu'),
    'conv1d_12_f20_even': ('conv_1d', (20, 4, 12, 1), 2, 18, vo.chain_bonds(12), 'tanh'),
}


def build_conv_case(name, cases=None, seed0=31):
  """Same quantities as build_rbm_case for the convolutional ansatz types
  (tests/golden/conv_small.npz, conv_wide.npz); `scale` = sum |last feature map|, the fp32 summation
  scale the logit tolerance is stated on."""
  cases = cases or CONV_CASES
  ansatz, geom, L, b, bonds, nonlin = cases[name]
  n = geom[2] * geom[3]
  rng = np.random.default_rng(seed0 + sorted(cases).index(name))
  theta = vo.conv_init_params(ansatz, geom, L, rng)
  theta = (theta + (0.03 if geom[0] <= 16 else 0.01) * rng.standard_normal(theta.size)).astype(np.float32)
  cfg = vo.random_configurations(n, b, np.random.RandomState(7))
  f64 = np.float64
  amp = lambda c: vo.ANSATZ[ansatz][0](theta, c, geom, L, nonlinearity=nonlin, dtype=f64)
  out = dict(theta=theta, configs=cfg, bonds=np.asarray(bonds, np.int32),
             shape=np.array(list(geom) + [L, b]), seed=np.array([SEED]), couplings=np.array([JX, JZ, BETA]))
  out['logit'], out['scale'] = vo.conv_forward(theta, cfg, ansatz, geom, L, nonlin, f64, return_tape='scale')
  out['eloc'] = vo.local_value(amp, cfg, bonds, JX, JZ, dtype=f64)
  u_sites, u_acc = vo.step_uniforms(SEED, np.arange(b), 3, n)
  i_up, i_dn = vo.propose_exchange(cfg, u_sites)
  _, acc, ratio = vo.mc_step(amp, cfg, i_up, i_dn, u_acc)
  out.update(i_up=i_up.astype(np.int32), i_dn=i_dn.astype(np.int32), u_accept=u_acc, accept=acc,
             ratio=ratio)
  acc_eg = vo.Accumulators(theta.size, f64)
  vo.energy_gradient_accumulate(acc_eg, theta, cfg, bonds, JX, JZ, -10.0, geom, L, f64, ansatz=ansatz,
                                nonlinearity=nonlin)
  out['eg_g1'], out['eg_g2'] = acc_eg.g1_total, acc_eg.g2_total
  out['eg_grad'] = vo.energy_gradient(acc_eg)
  return out


def _write(fname, cases, builder):
  data = {}
  for name in cases:
    for k, v in builder(name).items():
      data['{}/{}'.format(name, k)] = v
  path = os.path.join(HERE, fname)
  np.savez_compressed(path, **data)
  print('wrote', path, os.path.getsize(path), 'bytes')


def main():
  _write('conv_wide.npz', CONV_WIDE_CASES, lambda n: build_conv_case(n, CONV_WIDE_CASES, 51))
  _write('rbm_wide.npz', RBM_WIDE_CASES, lambda n: build_rbm_case(n, RBM_WIDE_CASES))
  if '--wide-only' in sys.argv:
    return
  conv = {}
  for name in CONV_CASES:
    for k, v in build_conv_case(name).items():
      conv['{}/{}'.format(name, k)] = v
  path = os.path.join(HERE, 'conv_small.npz')
  np.savez_compressed(path, **conv)
  print('wrote', path, os.path.getsize(path), 'bytes')
  if '--conv-only' in sys.argv:
    return
  rbm = {}
  for name in RBM_CASES:
    for k, v in build_rbm_case(name).items():
      rbm['{}/{}'.format(name, k)] = v
  path = os.path.join(HERE, 'rbm_small.npz')
  np.savez_compressed(path, **rbm)
  print('wrote', path, os.path.getsize(path), 'bytes')
  if '--rbm-only' in sys.argv:
    return
  data = {}
  for name in CASES:
    for k, v in build_case(name).items():
      data['{}/{}'.format(name, k)] = v
  path = os.path.join(HERE, 'vmc_small.npz')
  np.savez_compressed(path, **data)
  print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
  main()
