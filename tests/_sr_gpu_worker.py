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


def main():
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
