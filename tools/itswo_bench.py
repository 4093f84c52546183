"""LogOverlapImaginaryTimeSWO batch (training.py:756-761: one MC sweep, reset, accumulate with the
supervisor's local energy and the overlap ratio, Adam) at BASELINE config 3, through
vmc_epoch_log_overlap (one host call for all batches)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cgs_vmc_amd.engine import VmcEngine  # noqa: E402

n, h, L, b = 100, 256, 3, 4096
theta, cfg = bench.make_inputs(n, h, L, b, 0)
eng = VmcEngine(n, b, L, h)
eng.set_params(theta); eng.set_configs(cfg)
eng.set_bonds(bench.torus_bonds(10, 10, False), -1.0, 1.0)
eng.mc_steps(5 * n, want_accepted=False)
eng.epoch_log_overlap(0.12, 0, 3, n, 0.0, 1e-3, 0.9, 0.99, 1e-8)      # warm-up (also omega <- psi)
eng.synchronize()
K = 20
t0 = time.perf_counter()
e = eng.epoch_log_overlap(0.12, 0, K, n, 0.0, 1e-3, 0.9, 0.99, 1e-8)
eng.synchronize()
dt = (time.perf_counter() - t0) / K
print(json.dumps({'optimizer': 'LogOverlapITSWO', 'ms_per_batch': dt * 1e3,
                  'chain_evals_per_s': b / dt, 'energy_per_site': e / n}))
