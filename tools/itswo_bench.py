"""LogOverlapImaginaryTimeSWO batch (training.py:756-761: one MC sweep, reset, accumulate with the
supervisor's local energy and the overlap ratio, Adam) at BASELINE config 3: through
vmc_epoch_log_overlap (single rank) and through vmc_epoch_log_overlap_dist on a 1-rank RCCL
communicator created by the library (the multi-rank entry: in-stream ncclAllReduce of the 2P+8
accumulator floats between accumulate and Adam, no host round trip per batch)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cgs_vmc_amd import parallel  # noqa: E402
from cgs_vmc_amd.engine import VmcEngine  # noqa: E402

n, h, L, b = 100, 256, 3, 4096
theta, cfg = bench.make_inputs(n, h, L, b, 0)
K = 20
out = {'optimizer': 'LogOverlapITSWO', 'chains': b}
coll = parallel.rccl_collective(device=0, world=1, rank_=0)
for name in ('single_rank_entry', 'dist_entry_rccl_1_rank'):
  eng = VmcEngine(n, b, L, h)
  eng.set_params(theta); eng.set_configs(cfg)
  eng.set_bonds(bench.torus_bonds(10, 10, False), -1.0, 1.0)
  eng.mc_steps(5 * n, want_accepted=False)
  if name == 'single_rank_entry':
    run = lambda k: eng.epoch_log_overlap(0.12, 0, k, n, 0.0, 1e-3, 0.9, 0.99, 1e-8)
  else:
    run = lambda k: eng.epoch_log_overlap_dist(coll, 0.12, 0, k, n, 0.0, 1e-3, 0.9, 0.99, 1e-8)
  run(3)                                  # warm-up (also omega <- psi)
  eng.synchronize()
  t0 = time.perf_counter()
  e = run(K)
  eng.synchronize()
  dt = (time.perf_counter() - t0) / K
  out[name] = {'ms_per_batch': dt * 1e3, 'chain_evals_per_s': b / dt, 'energy_per_site': e / n}
  eng.close()
coll.close()
print(json.dumps(out))
