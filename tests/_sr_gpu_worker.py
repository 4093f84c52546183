"""Worker of tests/test_gpu_sr.py::test_sharded_sr_two_ranks_one_gpu: two ranks share one GPU
over gloo; each owns half of the chains, the accumulators and the SR matrix-vector buffer are
all-reduced, and both ranks must arrive at the dense fp64 solution over ALL samples."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist  # noqa: E402

from cgs_vmc_amd import parallel  # noqa: E402
from cgs_vmc_amd.engine import VmcEngine  # noqa: E402
from oracle import vmc_oracle as vo  # noqa: E402


def main_conv_general():
  """The same on the general convolution path (conv_1d with an 11-tap kernel): its matvec centres the per-sample
  weights on the mean of O_b . p over BOTH ranks (one more all-reduce between its two phases, vmc_sr_solve_dist)."""
  parallel.init_from_env('gloo')
  rank = parallel.rank()
  ansatz, n, L, f, k, b, n_store = 'conv_1d', 12, 2, 4, 11, 18, 2
  geom = (f, k, n, 1)
  rng = np.random.default_rng(0)
  theta = vo.conv_init_params(ansatz, geom, L, rng)
  theta += (0.03 * rng.standard_normal(theta.size)).astype(np.float32)
  bonds = vo.chain_bonds(n)
  local, offset = parallel.shard(b)
  eng = VmcEngine(n, local, L, f, nonlinearity='tanh', seed=2024, device=parallel.local_rank(), chain_offset=offset,
                  ansatz=ansatz, kernel_size=k, size_x=n, size_y=1)
  assert eng.kernel_path() == 6
  eng.set_params(theta)
  eng.set_bonds(bonds, -1.0, 1.0)
  eng.sr_reserve(n_store)
  eng.reset_accumulators()
  cfgs = []
  for j in range(n_store):
    cfg = vo.random_configurations(n, b, np.random.RandomState(20 + j))
    cfgs.append(cfg)
    eng.set_configs(cfg[offset:offset + local])
    eng.accumulate(0)
  parallel.allreduce_accumulators(eng)
  iters, res = parallel.sr_solve(eng, 0.01, 1e-6, 2000)
  x = eng.sr_get_solution()
  # both ranks hold the solution over ALL samples: the reference is the oracle's explicit S over the stored chains in
  # the order rank 0's, rank 1's of every batch -- S and f are sums over samples, the order does not matter
  cfg_all = np.concatenate(cfgs, 0)
  amp = lambda c: vo.ANSATZ[ansatz][0](theta, c, geom, L, nonlinearity='tanh', dtype=np.float64)
  e_all = vo.local_value(amp, cfg_all, bonds, -1.0, 1.0, dtype=np.float64)
  o = vo.ANSATZ[ansatz][2](theta, cfg_all, np.eye(cfg_all.shape[0]), geom, L, nonlinearity='tanh', dtype=np.float64)
  ref = vo.sr_solve(o, e_all, 0.01)
  oc = o - o.mean(0)
  err = np.abs(oc @ (x - ref)).max() / np.abs(oc @ ref).max()
  assert res <= 1e-4 and err <= 1e-2, (iters, res, err)
  # the op-by-op loop of parallel.sr_solve (two-phase matvec, round 6) arrives at the same solution
  os.environ['CGS_VMC_DIST_FUSED'] = '0'
  it2, res2 = parallel.sr_solve(eng, 0.01, 1e-6, 2000)
  os.environ.pop('CGS_VMC_DIST_FUSED')
  x2 = eng.sr_get_solution()
  assert abs(it2 - iters) <= 2 and np.abs(x2 - x).max() <= 1e-4 * np.abs(x).max(), (it2, iters, np.abs(x2 - x).max())
  eng.close()
  dist.barrier()
  dist.destroy_process_group()
  print('rank {} ok iters {} err {:.2e}'.format(rank, iters, err))


def main():
  if os.environ.get('CGS_SR_WORKER_CASE') == 'conv_general':
    return main_conv_general()
  parallel.init_from_env('gloo')
  rank = parallel.rank()
  n, h, L, b, n_store = 16, 32, 2, 64, 2
  rng = np.random.default_rng(0)
  theta = vo.init_params(n, h, L, rng)
  theta += (0.05 * rng.standard_normal(theta.size)).astype(np.float32)
  bonds = vo.torus_bonds(4, 4)
  local, offset = parallel.shard(b)
  eng = VmcEngine(n, local, L, h, seed=2024, device=parallel.local_rank(), chain_offset=offset)
  eng.set_params(theta)
  eng.set_bonds(bonds, -1.0, 1.0)
  eng.sr_reserve(n_store)
  eng.reset_accumulators()
  cfgs = []
  for k in range(n_store):
    cfg = vo.random_configurations(n, b, np.random.RandomState(20 + k))
    cfgs.append(cfg)
    eng.set_configs(cfg[offset:offset + local])
    eng.accumulate(0)
  parallel.allreduce_accumulators(eng)
  iters, res = parallel.sr_solve(eng, 0.01, 1e-6, 2000)
  x = eng.sr_get_solution()

  cfg_all = np.concatenate(cfgs, 0)
  amp = lambda c: vo.fc_psi(theta, c, h, L, -10.0, dtype=np.float64)
  e_all = vo.local_value(amp, cfg_all, bonds, -1.0, 1.0, dtype=np.float64)
  o = vo.per_sample_logit_grads(theta, cfg_all, h, L)
  ref = vo.sr_solve(o, e_all, 0.01)
  err = np.abs(x - ref).max() / np.abs(ref).max()
  assert res <= 1e-4 and err <= 2e-3, (iters, res, err)
  eng.close()
  dist.barrier()
  dist.destroy_process_group()
  print('rank {} ok iters {} err {:.2e}'.format(rank, iters, err))


if __name__ == '__main__':
  main()
