"""Worker of tests/test_parallel_gloo.py: world_size 2 and 8 (BASELINE configs 4 / 5 are 8-rank jobs)
check of the sharding + reduction logic of cgs_vmc_amd.parallel with the oracle standing in for the
GPU kernels."""
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


def _gather_rows(local_rows):
  """Global batch [B, N] from every rank's [B/W, N] shard."""
  import torch
  t = torch.from_numpy(np.ascontiguousarray(local_rows, np.float32))
  parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
  dist.all_gather(parts, t)
  return np.concatenate([p.numpy() for p in parts])


def routed_training_epochs(rank):
  """The PRODUCT routing for sharded chains -- training.run_optimization_epoch ->
  engine.epoch_*_dist(parallel.collective()) / parallel.sr_solve -> engine.sr_solve_dist, with
  the gloo host hook as transport -- against the unsharded oracle epoch.  The engine is the
  oracle-backed double of tests/oracle_engine.py (no GPU here); the same routing runs against
  libcgsvmc_hip.so in tests/test_gpu_dist.py."""
  os.environ.update(CGS_VMC_SEED='77', CGS_VMC_CONFIG_SEED='5', CGS_VMC_INIT_SEED='31')
  import cgs_vmc_amd.engine as engine_mod
  from cgs_vmc_amd import graph_builders, lattice, operators, session, training, utils, wavefunctions
  from tests.oracle_engine import OracleEngine
  engine_mod.VmcEngine = OracleEngine
  coll = parallel.collective()
  world = parallel.world_size()
  assert coll.world == world and coll.comm == 0 and coll.host_hook() is not None   # gloo: host hook
  assert coll.transport == 'host' and coll.device_hook() is None
  lb = 32 // world
  for name in ('LogOverlapITSWO', 'EnergyGradient', 'StochasticReconfiguration'):
    session.reset_default_graph()
    wavefunctions.reset_name_scope()
    hp = utils.create_hparams(wavefunction_type='fully_connected', num_sites=8, num_fc_layers=2,
                              fc_layer_size=16, batch_size=32, num_equilibration_sweeps=2,
                              num_monte_carlo_sweeps=1, num_batches_per_epoch=2,
                              learning_rates=[1e-2, 1e-3], learning_rate_stops=[1])
    n, h, L = hp.num_sites, hp.fc_layer_size, hp.num_fc_layers
    wf = wavefunctions.build_wavefunction(hp)
    ham = operators.HeisenbergHamiltonian(lattice.chain_bonds(n), -1.0, 1.0)
    opt = training.GROUND_STATE_OPTIMIZERS[name]()
    shared = {}
    ops = opt.build_opt_ops(wavefunction=wf, hamiltonian=ham, hparams=hp, shared_resources=shared)
    sess = session.Session()
    sess.run([session.global_variables_initializer(), session.local_variables_initializer()])
    cfg_var = shared[graph_builders.ResourceName.CONFIGS]
    assert isinstance(cfg_var._engine, OracleEngine) and cfg_var.local_batch == lb
    assert cfg_var.chain_offset == lb * rank
    # ---- unsharded oracle restatement of the same epochs on the gathered global batch
    theta = wf._get_theta().copy()
    cfg = _gather_rows(cfg_var.eval())
    bonds = ham._bonds_list
    sweeps = lambda th, c, k, s0: vo.run_sweeps(th, c, k, 77, s0, h, L, dtype=np.float64)[0]
    adam = vo.AdamState(theta.size)
    step = 0
    # entries whose gradient is identically zero up to rounding (b_out, always-active units) get
    # steps of order lr from Adam in ANY summation order: only well-conditioned ones are compared
    well = np.ones(theta.size, bool)
    for epoch in range(2):
      lr = vo.piecewise_constant(epoch, hp.learning_rate_stops, hp.learning_rates)
      cfg = sweeps(theta, cfg, hp.num_equilibration_sweeps * n, step); step += hp.num_equilibration_sweeps * n
      if name == 'LogOverlapITSWO':
        theta_w = theta.copy()
        for _ in range(hp.num_batches_per_epoch):
          cfg = sweeps(theta, cfg, n, step); step += n
          acc = vo.Accumulators(theta.size, np.float64)
          vo.log_overlap_accumulate(acc, theta, theta_w, cfg, bonds, -1.0, 1.0, -10.0, -10.0,
                                    hp.time_evolution_beta, h, L, np.float64)
          grad = vo.log_overlap_gradient(acc)
          well &= np.abs(grad) > 1e-3 * np.abs(grad).max()
          theta = vo.adam_apply(adam, theta, grad, lr, 0.9, hp.beta2, 1e-8)
      else:
        acc = vo.Accumulators(theta.size, np.float64)
        samples = []
        for _ in range(hp.num_batches_per_epoch):
          vo.energy_gradient_accumulate(acc, theta, cfg, bonds, -1.0, 1.0, -10.0, h, L, np.float64)
          samples.append(cfg.copy())
          cfg = sweeps(theta, cfg, n, step); step += n
        if name == 'EnergyGradient':
          grad = vo.energy_gradient(acc)
          well &= np.abs(grad) > 1e-3 * np.abs(grad).max()
          theta = vo.adam_apply(adam, theta, grad, lr, 0.9, hp.beta2, 1e-8)
        else:
          rows = np.concatenate(samples)
          o = vo.per_sample_logit_grads(theta, rows, h, L)
          amp = lambda c: vo.fc_psi(theta, c, h, L, -10.0, dtype=np.float64)
          e = vo.local_value(amp, rows, bonds, -1.0, 1.0, dtype=np.float64)
          theta = (theta - np.float32(lr) * vo.sr_solve(o, e, 0.01)).astype(np.float32)
      # ---- the routed, sharded epoch
      energy = opt.run_optimization_epoch(ops, sess, hp, epoch)
      assert abs(energy - acc.mean_energy()) < 1e-5 * max(1.0, abs(acc.mean_energy())), (name, energy)
      mine = cfg_var.eval()
      np.testing.assert_array_equal(mine, cfg[lb * rank:lb * (rank + 1)].astype(np.float32))
      # CG tolerance 1e-3 vs dense solve; Adam: fp32 accumulators summed over `world` shards (the 8-way
      # split re-associates more than the 2-way one: 2.1e-6 seen)
      tol = 2e-4 if name == 'StochasticReconfiguration' else (2e-6 if world <= 2 else 6e-6)
      got = wf._get_theta()
      err = np.abs(got - theta)[well].max()
      assert err <= tol * max(1.0, np.abs(theta).max()) and well.sum() > 0.8 * well.size, (name, epoch, err)
      theta = got.copy()    # keep the two trajectories on the same parameters epoch by epoch
      adam.m, adam.v = cfg_var._engine.adam.m.copy(), cfg_var._engine.adam.v.copy()
    # every rank holds the identical parameters
    every = _gather_rows(wf._get_theta()[None, :])
    for r in range(1, world):
      np.testing.assert_array_equal(every[0], every[r])


def main():
  parallel.init_from_env('gloo')
  world = int(os.environ['WORLD_SIZE'])
  assert parallel.is_distributed() and parallel.world_size() == world
  rank = parallel.rank()
  n, h, L, b = 8, 16, 2, 32
  rng = np.random.default_rng(0)
  theta = vo.init_params(n, h, L, rng)
  cfg = vo.random_configurations(n, b, np.random.RandomState(1))
  bonds = vo.chain_bonds(n)

  local, offset = parallel.shard(b)
  assert (local, offset) == (b // world, (b // world) * rank)
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

  assert parallel.allreduce_max(float(rank) + 0.5) == world - 0.5
  assert parallel.allreduce_sum(float(rank) + 1.0) == world * (world + 1) / 2
  routed_training_epochs(rank)
  dist.barrier()
  dist.destroy_process_group()
  print('rank {} ok'.format(rank))


if __name__ == '__main__':
  main()
