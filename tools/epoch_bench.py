"""Wall-clock of one EnergyGradient epoch (vmc_epoch_energy_gradient + Adam) at small,
launch-bound configurations, e.g. the reference's default hparams (utils.py:87-148).
Usage: python tools/epoch_bench.py [n_sites H L B n_batches]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cgs_vmc_amd.engine import VmcEngine  # noqa: E402

args = [int(x) for x in sys.argv[1:6]] if len(sys.argv) >= 6 else [40, 80, 3, 200, 50]
n, h, L, b, nb = args
theta, cfg = bench.make_inputs(n, h, L, b, 0)
eng = VmcEngine(n, b, L, h)
eng.set_params(theta); eng.set_configs(cfg)
bonds = [(i, (i + 1) % n) for i in range(n)]
eng.set_bonds(bonds, -1.0, 1.0)
for _ in range(3):
  eng.epoch_energy_gradient(100 * n, nb, n, 1e10)
  eng.apply_adam(0, 1e-3)
eng.synchronize()
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
  eng.epoch_energy_gradient(100 * n, nb, n, 1e10)
  eng.apply_adam(0, 1e-3)
eng.synchronize()
dt = (time.perf_counter() - t0) / reps
print(json.dumps({'config': args, 'epoch_ms': dt * 1e3,
                  'launch_equiv_us_per_batch': dt * 1e6 / nb}))
