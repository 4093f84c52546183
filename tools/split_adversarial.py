"""EXPERIMENT bound: the 3 x bf16 split kernels (CGS_VMC_SPLIT_BF16=2: row kernel k_tail16r and
sampler k_sweep16s) against the native fp32 kernels AND the fp64 oracle on operands chosen to hurt:
  range    every weight row spans 2^20 in magnitude (sign * 2^-U(0, 20), rows rescaled to the Sonnet norm)
  cancel   hidden units in exactly cancelling pairs (equal incoming rows, opposite outgoing weights), so every layer's
           pre-activation is a sum of terms that cancel to rounding
  trained  the weights after N epochs of EnergyGradient + Adam on the 10 x 10 torus (native kernels), i.e. what the
           kernels see in production rather than at initialisation
  init     truncated-normal initialisation (the case tests/test_gpu_split.py already covers), for scale
Per case: max |logit - fp64| / max(1, |logit|), max |E_loc - fp64| / max(1, |E_loc|) for both kernels, their ratio,
and the Metropolis decisions of 3 injected steps that differ from the fp64 decision outside the band
|ratio - sqrt(u)| < 1e-4 ratio (the band tests/test_gpu_engine.py allows the native kernel).
  python tools/split_adversarial.py [epochs]        prints a table; --json for one JSON line"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vmc_oracle as vo  # noqa: E402  (a checker here: this is a test tool, not the product path)

N, H, L, B = 100, 256, 3, 256
BONDS = vo.torus_bonds(10, 10)


def layout(theta):
  """views (W1 [N,H], b1, [(W [H,H], b)], w_out [H], b_out) into a copy of theta (order: wavefunctions.py:167-175)"""
  t = theta.copy()
  o = 0
  def take(shape):
    nonlocal o
    n = int(np.prod(shape)); v = t[o:o + n].reshape(shape); o += n
    return v
  w1 = take((N, H)); b1 = take((H,))
  hh = [(take((H, H)), take((H,))) for _ in range(L - 1)]
  wo = take((H, 1)); bo = take((1,))
  assert o == t.size
  return t, w1, b1, hh, wo, bo


def make_range(rng):
  theta = vo.init_params(N, H, L, rng)
  t, w1, b1, hh, wo, bo = layout(theta)
  for w in [w1] + [x[0] for x in hh] + [wo]:
    mag = np.exp2(-rng.uniform(0.0, 20.0, w.shape)).astype(np.float32)
    sgn = rng.choice(np.float32([-1.0, 1.0]), w.shape)
    v = sgn * mag
    v *= (1.0 / np.sqrt(w.shape[0])) / np.sqrt((v.astype(np.float64) ** 2).mean()).astype(np.float32)   # Sonnet's sigma = 1/sqrt(fan_in)
    w[...] = v
  for b in [b1] + [x[1] for x in hh]:
    b[...] = (0.1 * rng.standard_normal(b.shape)).astype(np.float32)
  return t


def make_cancel(rng):
  theta = vo.init_params(N, H, L, rng)
  t, w1, b1, hh, wo, bo = layout(theta)
  w1 *= 3.0                                       # large activations: what cancels is big
  w1[:, 1::2] = w1[:, 0::2]; b1[1::2] = b1[0::2]  # units 2k, 2k+1 of layer 1 are twins ...
  for (w, b) in hh:
    w *= 3.0
    w[1::2, :] = -w[0::2, :]                      # ... with opposite outgoing weights: their contributions cancel exactly
    w[:, 1::2] = w[:, 0::2]; b[1::2] = b[0::2]    # and the next layer's units are twins again
    b[...] = b + np.float32(0.05)                 # (a positive bias keeps the relu open: the logit is bias-driven)
  wo[1::2] = -wo[0::2]
  return t


def make_trained(epochs):
  from cgs_vmc_amd.engine import VmcEngine
  os.environ.pop('CGS_VMC_SPLIT_BF16', None)
  rng = np.random.default_rng(3)
  theta = vo.init_params(N, H, L, rng)
  eng = VmcEngine(N, 1024, L, H, seed=11)
  eng.set_params(theta)
  eng.set_configs(vo.random_configurations(N, 1024, np.random.RandomState(4)))
  eng.set_bonds(BONDS, -1.0, 1.0)
  e = None
  for ep in range(epochs):
    eng.epoch_energy_gradient(10 * N if ep == 0 else 2 * N, 8, N, 1e10)
    e = eng.apply_adam(0, 1e-3 if ep < epochs // 2 else 3e-4)
  out = eng.get_params()
  eng.close()
  return out, e


def evaluate(theta, split):
  from cgs_vmc_amd.engine import VmcEngine
  if split:
    os.environ['CGS_VMC_SPLIT_BF16'] = '2'
  else:
    os.environ.pop('CGS_VMC_SPLIT_BF16', None)
  cfg = vo.random_configurations(N, B, np.random.RandomState(7))
  eng = VmcEngine(N, B, L, H, seed=2024)
  assert eng.kernel_path() == (5 if split else 0)
  eng.set_params(theta); eng.set_configs(cfg); eng.set_bonds(BONDS, -1.0, 1.0)
  logit = eng.amplitude()[0].astype(np.float64)
  eloc = eng.local_energy()[0].astype(np.float64)
  amp = lambda c: vo.fc_psi(theta, c, H, L, dtype=np.float64)
  masks = []
  cur = cfg
  flips = 0
  decided = 0
  for step in range(3):
    u_sites, u_acc = vo.step_uniforms(99, np.arange(B), step, N)
    i_up, i_dn = vo.propose_exchange(cur, u_sites)
    new_ref, acc_ref, ratios = vo.mc_step(amp, cur, i_up, i_dn, u_acc)
    mask = eng.mc_step_injected(i_up, i_dn, u_acc)
    band = np.abs(ratios - np.sqrt(u_acc.astype(np.float64))) < 1e-4 * np.maximum(ratios, 1e-30)
    flips += int((mask[~band] != acc_ref[~band]).sum())
    decided += int((~band).sum())
    cur = eng.get_configs()
  eng.close()
  ref_logit = vo.fc_logit(theta, cfg, H, L, dtype=np.float64)
  ref_eloc = vo.local_value(amp, cfg, BONDS, -1.0, 1.0, dtype=np.float64)
  return dict(logit_err=float(np.abs(logit - ref_logit).max() / max(1.0, np.abs(ref_logit).max())),
              eloc_err=float(np.abs(eloc - ref_eloc).max() / max(1.0, np.abs(ref_eloc).max())),
              logit_scale=float(np.abs(ref_logit).max()), eloc_scale=float(np.abs(ref_eloc).max()),
              wrong_decisions=flips, decisions=decided), logit, eloc


def main():
  epochs = int([a for a in sys.argv[1:] if a.isdigit()][0]) if [a for a in sys.argv[1:] if a.isdigit()] else 60
  rng = np.random.default_rng(1)
  cases = {'init': vo.init_params(N, H, L, rng) + (0.03 * rng.standard_normal(vo.num_params(N, H, L))).astype(np.float32),
           'range': make_range(rng), 'cancel': make_cancel(rng)}
  trained, e = make_trained(epochs)
  cases['trained'] = trained
  out = {'shape': [N, H, L, B], 'train_epochs': epochs, 'trained_energy_per_site': None if e is None else e / N, 'cases': {}}
  worst = 0.0
  for name, theta in cases.items():
    nat, ln, en = evaluate(theta, False)
    spl, ls, es = evaluate(theta, True)
    floor_l, floor_e = 1e-7, 1e-6            # errors below these are rounding of the comparison itself
    r_l = max(spl['logit_err'], floor_l) / max(nat['logit_err'], floor_l)
    r_e = max(spl['eloc_err'], floor_e) / max(nat['eloc_err'], floor_e)
    worst = max(worst, r_l, r_e)
    out['cases'][name] = dict(native=nat, split=spl, ratio_logit=r_l, ratio_eloc=r_e,
                              split_vs_native_logit=float(np.abs(ls - ln).max() / max(1.0, np.abs(ln).max())),
                              split_vs_native_eloc=float(np.abs(es - en).max() / max(1.0, np.abs(en).max())))
  out['worst_ratio_split_over_native'] = worst
  if '--json' in sys.argv:
    print(json.dumps(out))
    return
  print('10 x 10 torus, FC 3 x 256, {} chains; split = CGS_VMC_SPLIT_BF16=2 (k_tail16r + k_sweep16s); errors / max(1, scale) against the fp64 oracle'.format(B))
  print('{:<9s}{:>12s}{:>12s}{:>8s}{:>12s}{:>12s}{:>8s}{:>16s}{:>16s}'.format(
      'case', 'logit nat', 'logit split', 'ratio', 'E_loc nat', 'E_loc split', 'ratio', 'wrong dec. nat', 'wrong dec. split'))
  for name, c in out['cases'].items():
    print('{:<9s}{:>12.2e}{:>12.2e}{:>8.2f}{:>12.2e}{:>12.2e}{:>8.2f}{:>10d}/{:<5d}{:>10d}/{:<5d}'.format(
        name, c['native']['logit_err'], c['split']['logit_err'], c['ratio_logit'], c['native']['eloc_err'], c['split']['eloc_err'],
        c['ratio_eloc'], c['native']['wrong_decisions'], c['native']['decisions'], c['split']['wrong_decisions'], c['split']['decisions']))
  print('scales: ' + ', '.join('{} |logit| {:.1f} |E_loc| {:.1f}'.format(k, c['native']['logit_scale'], c['native']['eloc_scale']) for k, c in out['cases'].items()))
  print('trained: {} epochs, energy per site {:.4f}'.format(epochs, out['trained_energy_per_site']))
  print('worst split-error / native-error: {:.2f}'.format(worst))


if __name__ == '__main__':
  main()
