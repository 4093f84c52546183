"""Worker of tests/test_parallel_gloo.py: world_size-2 check of the sharding + reduction
logic of cgs_vmc_amd.parallel with the oracle standing in for the GPU kernels."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist  # noqa: E402

from cgs_vmc_amd import parallel  # noqa: E402
from oracle import vmc_oracle as vo  # noqa: E402


class _OracleSrEngine:
  """numpy stand-in for VmcEngine's SR entry points (the recurrence of csrc/sr.hip in fp64) so
  that parallel.sr_solve's sharded CG loop runs without a GPU."""

  def __init__(self, o_local, n_total, f, o_mean):
    self.o, self.n, self.f, self.o_mean = o_local, n_total, f, o_mean

  def sr_begin(self):
    self.x = np.zeros_like(self.f); self.r = self.f.copy(); self.p = self.f.copy()
    self.rr = float(self.r @ self.r)
    return self.rr

  def sr_matvec_partial(self):
    t = self.o @ self.p
    self.buf = np.concatenate([self.o.T @ t, [t.sum()]])

  def sr_get_buffer(self):
    return self.buf.astype(np.float32)

  def sr_set_buffer(self, buf):
    self.buf = np.asarray(buf, np.float64)

  def sr_cg_update(self, lam):
    q = self.buf[:-1] / self.n - self.o_mean * (self.buf[-1] / self.n) + lam * self.p
    alpha = self.rr / float(self.p @ q)
    self.x += alpha * self.p
    self.r -= alpha * q
    rr_new = float(self.r @ self.r)
    self.p = self.r + (rr_new / self.rr) * self.p
    self.rr = rr_new
    return rr_new


def main():
  parallel.init_from_env('gloo')
  assert parallel.is_distributed() and parallel.world_size() == 2
  rank = parallel.rank()
  n, h, L, b = 8, 16, 2, 32
  rng = np.random.default_rng(0)
  theta = vo.init_params(n, h, L, rng)
  cfg = vo.random_configurations(n, b, np.random.RandomState(1))
  bonds = vo.chain_bonds(n)

  local, offset = parallel.shard(b)
  assert (local, offset) == (16, 16 * rank)
  try:
    parallel.shard(33)
    raise AssertionError('expected ValueError')
  except ValueError:
    pass

  # sampling keyed by GLOBAL chain id: a shard walks exactly the chains of the full batch
  full, _ = vo.run_sweeps(theta, cfg, 12, 7, 0, h, L, chain_offset=0, dtype=np.float64)
  mine, _ = vo.run_sweeps(theta, cfg[offset:offset + local], 12, 7, 0, h, L,
                          chain_offset=offset, dtype=np.float64)
  np.testing.assert_array_equal(mine, full[offset:offset + local])

  # accumulators: sum over ranks == unsharded
  def pack(acc):
    return np.concatenate([acc.g1_total, acc.g2_total,
                           [acc.e_total, acc.e_count, acc.r_total, acc.r_count, acc.g_count, 0, 0, 0]]
                          ).astype(np.float32)
  acc_full = vo.Accumulators(theta.size, np.float64)
  vo.energy_gradient_accumulate(acc_full, theta, full, bonds, -1.0, 1.0, -10.0, h, L, np.float64)
  acc_mine = vo.Accumulators(theta.size, np.float64)
  vo.energy_gradient_accumulate(acc_mine, theta, mine, bonds, -1.0, 1.0, -10.0, h, L, np.float64)
  red = parallel.reduce_accumulators_host(pack(acc_mine))
  ref = pack(acc_full)
  p = theta.size
  np.testing.assert_allclose(red[:2 * p], ref[:2 * p], rtol=1e-4, atol=1e-4)
  assert abs(red[2 * p] - ref[2 * p]) < 1e-3 and red[2 * p + 1] == b
  # mean_tensor count stays the number of accumulate CALLS, so sharded == unsharded gradient
  assert red[2 * p + 4] == 1
  g_red = red[p:2 * p] / red[2 * p + 4] - (red[2 * p] / red[2 * p + 1]) * red[:p] / red[2 * p + 4]
  g_ref = vo.energy_gradient(acc_full)
  np.testing.assert_allclose(g_red, g_ref, rtol=2e-3, atol=2e-4)

  # stochastic reconfiguration: samples sharded by chain, one P+1 all-reduce per CG iteration
  o_full = vo.per_sample_logit_grads(theta, full, h, L)
  e_full = vo.local_value(lambda c: vo.fc_psi(theta, c, h, L, -10.0, dtype=np.float64), full,
                          bonds, -1.0, 1.0, dtype=np.float64)
  _, f = vo.sr_system(o_full, e_full)
  fake = _OracleSrEngine(o_full[offset:offset + local], b, f, o_full.mean(0))
  iters, res = parallel.sr_solve(fake, 0.01, 1e-5, 500)
  x_ref = vo.sr_solve(o_full, e_full, 0.01)
  assert res <= 1e-5 and 0 < iters < 500, (iters, res)
  assert np.abs(fake.x - x_ref).max() <= 1e-3 * np.abs(x_ref).max(), np.abs(fake.x - x_ref).max()

  assert parallel.allreduce_max(float(rank) + 0.5) == 1.5
  assert parallel.allreduce_sum(float(rank) + 1.0) == 3.0
  dist.barrier()
  dist.destroy_process_group()
  print('rank {} ok'.format(rank))


if __name__ == '__main__':
  main()
