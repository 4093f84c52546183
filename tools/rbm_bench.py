"""Step time of the RestrictedBoltzmannNetwork ansatz at the config-3 shape (10x10 torus, H = 256,
num_fc_layers = 2 relu layers + the cosh layer = the same two H x H products per amplitude as the
3x256 fully-connected benchmark network, 4096 chains).  Usage: python tools/rbm_bench.py [L]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cgs_vmc_amd.engine import VmcEngine  # noqa: E402
from oracle import vmc_oracle as vo  # noqa: E402  (parameter initialisation only)

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n, h, b = 100, 256, 4096
rng = np.random.default_rng(1234)
theta = vo.rbm_init_params(n, h, L, rng)
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
