"""Step time of the RestrictedBoltzmannNetwork ansatz at the config-3 shape (10x10 torus, H = 256,
num_fc_layers = 2 relu layers + the cosh layer = the same two H x H products per amplitude as the
3x256 fully-connected benchmark network, 4096 chains).  Usage: python tools/rbm_bench.py [L [H]]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cgs_vmc_amd.engine import VmcEngine  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n, h, b = 100, 256, 4096
H = int(sys.argv[2]) if len(sys.argv) > 2 else h
h = H


def rbm_init(n_sites, units, layers, rng):
  """Sonnet-default init in the parameter order of include/cgsvmc.h (onsite layer first):
  w ~ truncated normal with sigma = 1/sqrt(fan_in), b = 0."""
  def tn(shape, fan_in):
    w = rng.standard_normal(shape)
    bad = np.abs(w) > 2
    while bad.any():
      w[bad] = rng.standard_normal(int(bad.sum()))
      bad = np.abs(w) > 2
    return (w / np.sqrt(fan_in)).ravel()
  parts = [tn((n_sites, 1), n_sites), np.zeros(1), tn((n_sites, units), n_sites), np.zeros(units)]
  for _ in range(layers):
    parts += [tn((units, units), units), np.zeros(units)]
  return np.concatenate(parts).astype(np.float32)


theta = rbm_init(n, h, L, np.random.default_rng(1234))
_, cfg = bench.make_inputs(n, h, 3, b, 0)
eng = VmcEngine(n, b, L, h, ansatz='rbm')
eng.set_params(theta); eng.set_configs(cfg)
eng.set_bonds(bench.torus_bonds(10, 10, False), -1.0, 1.0)
for _ in range(5):
  eng.mc_steps(n, want_accepted=False)


def step():
  eng.reset_accumulators(); eng.accumulate(0); eng.mc_steps(n, want_accepted=False)


for _ in range(3):
  step()
eng.timing_enable(2); eng.timing_reset(); eng.synchronize()
t0 = time.perf_counter()
K = 20
for _ in range(K):
  step()
eng.synchronize()
dt = (time.perf_counter() - t0) / K
out = {'ansatz': 'rbm', 'num_fc_layers': L, 'ms_per_step': dt * 1e3, 'chain_evals_per_s': b / dt}
for name in ('sweep', 'tail_eloc'):
  ms, cnt = eng.timing_get(name)
  out[name + '_ms'] = ms / max(cnt, 1)
print(json.dumps(out))
