"""LogOverlapImaginaryTimeSWO batch (training.py:756-761: one MC sweep, reset, accumulate with the
supervisor's local energy and the overlap ratio, Adam) at BASELINE config 3.

  python tools/itswo_bench.py                 one GPU: vmc_epoch_log_overlap against vmc_epoch_log_overlap_dist
                                              on a 1-rank RCCL communicator created by the library
  python tools/itswo_bench.py --gpus N        N ranks (started from here, or by torch.distributed.run), 4096
                                              chains each: the multi-rank entry the training loop uses, the
                                              all-reduce of the 2P+8 accumulator floats issued IN STREAM
                                              between accumulate and Adam by the library (transport:
                                              CGS_VMC_TRANSPORT = torch | rccl | host; backend:
                                              CGS_VMC_DIST_BACKEND, gloo lets the ranks share one GPU)
Prints one JSON line (rank 0): ms per batch = max over ranks."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--gpus', type=int, default=1)
ap.add_argument('--batches', type=int, default=20)
ap.add_argument('--chains', type=int, default=4096, help='per rank')
args = ap.parse_args()
if args.gpus > 1 and os.environ.get('WORLD_SIZE') is None:
  sys.exit(bench.spawn_ranks(args.gpus, __file__))       # before any GPU call

from cgs_vmc_amd import parallel  # noqa: E402
from cgs_vmc_amd.engine import VmcEngine  # noqa: E402

n, h, L, b = 100, 256, 3, args.chains
K = args.batches
world = int(os.environ.get('WORLD_SIZE', '1'))
if world > 1:
  parallel.init_from_env('nccl')
rank, dev = parallel.rank(), parallel.local_rank()
theta, cfg = bench.make_inputs(n, h, L, b, rank * b)
out = {'optimizer': 'LogOverlapITSWO', 'chains_per_rank': b, 'ranks': world}


def timed(eng, run):
  run(3)                                  # warm-up (also omega <- psi)
  eng.synchronize()
  if world > 1:
    import torch.distributed as dist
    dist.barrier()
  t0 = time.perf_counter()
  e = run(K)
  eng.synchronize()
  dt = (time.perf_counter() - t0) / K
  if world > 1:
    dt = parallel.allreduce_max(dt)
  return {'ms_per_batch': dt * 1e3, 'chain_evals_per_s': world * b / dt, 'energy_per_site': e / n}


def engine():
  eng = VmcEngine(n, b, L, h, device=dev, chain_offset=rank * b)
  eng.set_params(theta); eng.set_configs(cfg)
  eng.set_bonds(bench.torus_bonds(10, 10, False), -1.0, 1.0)
  eng.mc_steps(5 * n, want_accepted=False)
  return eng


if world == 1:
  coll = parallel.rccl_collective(device=0, world=1, rank_=0)
  for name in ('single_rank_entry', 'dist_entry_rccl_1_rank'):
    eng = engine()
    if name == 'single_rank_entry':
      run = lambda k: eng.epoch_log_overlap(0.12, 0, k, n, 0.0, 1e-3, 0.9, 0.99, 1e-8)
    else:
      run = lambda k: eng.epoch_log_overlap_dist(coll, 0.12, 0, k, n, 0.0, 1e-3, 0.9, 0.99, 1e-8)
    out[name] = timed(eng, run)
    eng.close()
  coll.close()
else:
  import torch.distributed as dist
  coll = parallel.collective()
  out.update(backend=dist.get_backend(), transport=coll.transport, allreduce_floats=2 * theta.size + 8)
  eng = engine()
  out['dist_entry'] = timed(eng, lambda k: eng.epoch_log_overlap_dist(coll, 0.12, 0, k, n, 0.0, 1e-3, 0.9, 0.99, 1e-8))
  # every rank took the identical Adam steps
  import numpy as np
  th = eng.get_params()
  out['theta_identical_on_all_ranks'] = bool(parallel.allreduce_max(float(np.abs(th).sum())) ==
                                             -parallel.allreduce_max(-float(np.abs(th).sum())))
  eng.close()
  dist.barrier()
  dist.destroy_process_group()
if rank == 0:
  print(json.dumps(out))
