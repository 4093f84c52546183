"""Times the stochastic-reconfiguration solve at BASELINE config 3 (10x10 torus, FC 3x256,
4096 chains per rank): one epoch slice of `n_store` accumulate calls, then CG iterations.

  python tools/sr_bench.py [n_store] [cg_iters] [conv]      one GPU   (conv: the 5 x 16-filter k 5 network)
  python tools/sr_bench.py [n_store] [cg_iters] --gpus N    N ranks (started from here or by
      torch.distributed.run): the stored samples are sharded, vmc_sr_solve_dist issues the P+1-float
      all-reduce of every CG iteration in stream (transport: CGS_VMC_TRANSPORT; backend:
      CGS_VMC_DIST_BACKEND, gloo lets the ranks share one GPU); ms per iteration = max over ranks."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

argv = [a for a in sys.argv[1:]]
gpus = 1
if '--gpus' in argv:
  i = argv.index('--gpus')
  gpus = int(argv[i + 1])
  del argv[i:i + 2]
if gpus > 1 and os.environ.get('WORLD_SIZE') is None:
  sys.exit(bench.spawn_ranks(gpus, __file__))            # before any GPU call

from cgs_vmc_amd import parallel  # noqa: E402
from cgs_vmc_amd.engine import VmcEngine  # noqa: E402

n_store = int(argv[0]) if len(argv) > 0 else 8
iters = int(argv[1]) if len(argv) > 1 else 20
conv = len(argv) > 2 and argv[2] == 'conv'
world = int(os.environ.get('WORLD_SIZE', '1'))
if world > 1:
  parallel.init_from_env('nccl')
rank, dev = parallel.rank(), parallel.local_rank()
n, h, L, b = 100, 256, 3, 4096
if conv:
  h, L = 16, 5
  theta, cfg = bench.make_inputs(n, h, L, b, rank * b, 'conv_2d', 5)
  eng = VmcEngine(n, b, L, h, device=dev, chain_offset=rank * b, ansatz='conv_2d', kernel_size=5, size_x=10, size_y=10)
else:
  theta, cfg = bench.make_inputs(n, h, L, b, rank * b)
  eng = VmcEngine(n, b, L, h, device=dev, chain_offset=rank * b)
eng.set_params(theta); eng.set_configs(cfg)
eng.set_bonds(bench.torus_bonds(10, 10, False), -1.0, 1.0)
eng.sr_reserve(n_store)
eng.mc_steps(10 * n)
coll = parallel.collective() if world > 1 else None
if world > 1:
  eng.epoch_energy_gradient_dist(coll, 0, n_store, n, 0.0)     # accumulators come back all-reduced
  solve = lambda k: eng.sr_solve_dist(coll, 0.01, 0.0, k)
else:
  eng.reset_accumulators()
  for _ in range(n_store):
    eng.accumulate(0)
    eng.mc_steps(n)
  solve = lambda k: eng.sr_solve(0.01, 0.0, k)
solve(3)          # warm-up
eng.timing_enable(True); eng.timing_reset()
eng.synchronize()
if world > 1:
  import torch.distributed as dist
  dist.barrier()
t0 = time.perf_counter()
it, res = solve(iters)
eng.synchronize(); t1 = time.perf_counter()
wall = t1 - t0
if world > 1:
  wall = parallel.allreduce_max(wall)
ms, launches = eng.timing_get('sr_matvec')
f_amp = 2 * (n * h + (L - 1) * h * h + h) if not conv else 2 * n * 25 * (h + (L - 1) * h * h)
# per stored sample and CG iteration the reverse-mode form executes 2 F_amp (t_b = sum_l delta_l .
# (a_{l-1} V_l): 1 F_amp; u = sum_b t_b O_b: 1 F_amp); the forward-mode tangent chain of round 1
# needed 3 F_amp, which is the count its 50 TFLOP/s figure was quoted on
flops = 2.0 * f_amp * b * n_store
out = {
    'n_store': n_store, 'samples': world * n_store * b, 'ranks': world, 'cg_iters': it, 'rel_residual': res,
    'wall_ms_per_iter': wall * 1e3 / max(it, 1),
    'matvec_ms': ms / max(launches, 1), 'matvec_ms_per_batch': ms / max(launches, 1) / n_store,
    'matvec_tflops_executed': flops / (ms / max(launches, 1) * 1e-3) / 1e12,
    'matvec_frac_of_fp32_mfma_peak': flops / (ms / max(launches, 1) * 1e-3) / 1e12 / 157.3,
    'matvec_tflops_round1_count': 1.5 * flops / (ms / max(launches, 1) * 1e-3) / 1e12,
}
if world > 1:
  out.update(backend=dist.get_backend(), transport=coll.transport, allreduce_floats_per_iter=int(theta.size) + 1)
  eng.close()
  dist.barrier()
  dist.destroy_process_group()
if rank == 0:
  print(json.dumps(out))
