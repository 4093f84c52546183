"""GPU: BASELINE config 4 (10x10, FC 3x256, 32,768 chains over 8 ranks) and config 5 (16x16 J1-J2,
FC 6x256, 8,192 chains over 8 ranks) as EIGHT-WAY jobs on the one GPU a box has.

The batch is the only parallel axis of the reference (graph_builders.py:57-88: no cross-row op but
the final reduce_sum), so an 8-rank job is eight engines with chain_offset = r B/8 plus the
accumulator all-reduce.  Two forms:
 (a) the eight shards one after the other against ONE engine holding the whole batch: chains
     bit-identical per shard after two sweeps, the eight accumulator buffers add up to the unsharded
     one, and the library's own world_size = 8 reduction (g_count / 8) applied to them;
 (b) eight ranks AT ONCE -- eight threads of this process, one vmc_ctx each, meeting in the host
     all-reduce hook (tests/rank_threads.py; the pool allows at most 6 processes on a GPU, so eight
     gloo rank processes cannot share it) -- through vmc_epoch_energy_gradient_dist /
     vmc_epoch_log_overlap_dist / vmc_sr_solve_dist / vmc_update_norm_dist / vmc_evaluate, against the
     unsharded epoch of one engine.
The product routing through training.run_optimization_epoch runs with 4 gloo rank processes on the
GPU (tests/test_gpu_dist.py) and with 8 on the CPU (tests/test_parallel_gloo.py).
"""
import numpy as np
import pytest

from oracle import vmc_oracle as vo
from tests.rank_threads import run_ranks

pytestmark = pytest.mark.gpu

WORLD = 8
# name: (lx, ly, next-nearest bonds too, L, H, GLOBAL chains)
JOBS = {
    'config4_10x10_fc3x256_b32768_8ranks': (10, 10, False, 3, 256, 32768),
    'config5_16x16j1j2_fc6x256_b8192_8ranks': (16, 16, True, 6, 256, 8192),
}


def _inputs(name):
  import bench
  lx, ly, nnn, L, h, b = JOBS[name]
  n = lx * ly
  theta, cfg = bench.make_inputs(n, h, L, b, 0)
  bonds = vo.torus_bonds(lx, ly, nnn)
  jx, jz = bench.couplings(len(bonds), nnn)
  return (n, h, L, b), theta, cfg, bonds, jx, jz


def _engine(shape, theta, cfg, bonds, jx, jz, offset=0, chains=None, seed=2024):
  from cgs_vmc_amd.engine import VmcEngine
  n, h, L, b = shape
  rows = cfg if chains is None else cfg[offset:offset + chains]
  eng = VmcEngine(n, len(rows), L, h, seed=seed, chain_offset=offset)
  eng.set_params(theta)
  eng.set_configs(rows)
  eng.set_bonds(bonds, jx, jz)
  return eng


@pytest.mark.parametrize('name', sorted(JOBS))
def test_eight_shards_one_after_the_other_equal_the_whole_batch(name):
  from cgs_vmc_amd import _hip, parallel
  shape, theta, cfg, bonds, jx, jz = _inputs(name)
  n, h, L, b = shape
  p = theta.size
  lb = b // WORLD
  whole = _engine(shape, theta, cfg, bonds, jx, jz)
  whole.reset_accumulators()
  whole.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  acc_whole = whole.get_accumulators().astype(np.float64)
  e_whole = whole.local_energy()[0]
  whole.mc_steps(2 * n)
  out_whole = whole.get_configs()
  whole.close()
  assert (out_whole != cfg).any() and (out_whole.sum(1) == cfg.sum(1)).all()
  parts = []
  for r in range(WORLD):
    sh = _engine(shape, theta, cfg, bonds, jx, jz, offset=r * lb, chains=lb)
    sh.reset_accumulators()
    sh.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
    parts.append(sh.get_accumulators())
    np.testing.assert_array_equal(sh.local_energy()[0], e_whole[r * lb:(r + 1) * lb], err_msg='rank %d' % r)
    sh.mc_steps(2 * n)
    np.testing.assert_array_equal(sh.get_configs(), out_whole[r * lb:(r + 1) * lb], err_msg='rank %d' % r)
    if r < WORLD - 1:
      sh.close()
  # sum over the eight ranks == the unsharded accumulators (fp32 re-association only)
  tot = np.sum([q.astype(np.float64) for q in parts], axis=0)
  scale = np.abs(acc_whole[:2 * p]).max()
  assert np.abs(tot[:2 * p] - acc_whole[:2 * p]).max() < 1e-4 * scale
  assert abs(tot[2 * p] - acc_whole[2 * p]) < 1e-4 * abs(acc_whole[2 * p])
  assert tot[2 * p + 1] == b and tot[2 * p + 4] == WORLD and acc_whole[2 * p + 4] == 1
  # the library's own reduction at world_size = 8 on the last shard's ctx: the "other seven ranks" are
  # the buffers collected above, folded in rank order by the host hook

  class Eight(parallel.Collective):
    def allreduce_host(self, buf, op='sum'):
      assert op == 'sum' and buf.size == 2 * p + 8
      np.testing.assert_array_equal(buf, parts[WORLD - 1])
      out = parts[0].copy()
      for q in parts[1:]:
        out = out + q
      buf[...] = out
      return buf

  sh.allreduce_accumulators_dist(Eight(0, WORLD, use_hook=True))
  red = sh.get_accumulators().astype(np.float64)
  assert red[2 * p + 4] == 1.0                                  # g_count: accumulate CALLS, / 8 again
  assert red[2 * p + 1] == b
  np.testing.assert_allclose(red[:2 * p], tot[:2 * p], rtol=0, atol=2e-6 * scale)   # fp32 fold vs fp64 sum
  g_red = sh.get_gradient(_hip.VMC_MODE_ENERGY_GRADIENT).astype(np.float64)
  g_ref = acc_whole[p:2 * p] - (acc_whole[2 * p] / acc_whole[2 * p + 1]) * acc_whole[:p]
  e_mean = abs(acc_whole[2 * p] / acc_whole[2 * p + 1])
  assert np.abs(g_red - g_ref).max() < 1e-5 * scale * (1.0 + e_mean)     # g = g2 - <E> g1: fp32 sums, then a difference
  sh.close()


def _small():
  n, h, L, b = 16, 32, 2, 128
  rng = np.random.default_rng(3)
  theta = vo.init_params(n, h, L, rng) + (0.05 * rng.standard_normal(vo.num_params(n, h, L))).astype(np.float32)
  cfg = vo.random_configurations(n, b, np.random.RandomState(4))
  return (n, h, L, b), theta, cfg, vo.torus_bonds(4, 4), -1.0, 1.0


def _well(g):
  return np.abs(g) > 1e-3 * np.abs(g).max()


def test_eight_rank_threads_energy_gradient_epoch_and_adam():
  """training.py:608-622 on 8 ranks: vmc_epoch_energy_gradient_dist (update_norm MAX + accumulator SUM
  all-reduces) then the identical Adam step on every rank."""
  from cgs_vmc_amd import _hip
  shape, theta, cfg, bonds, jx, jz = _small()
  n, h, L, b = shape
  lb = b // WORLD
  ref = _engine(shape, theta, cfg, bonds, jx, jz, seed=11)
  ref.set_shift(-40.0)
  ref.epoch_energy_gradient(2 * n, 3, n, 1e10)
  g_ref = ref.get_gradient(_hip.VMC_MODE_ENERGY_GRADIENT)
  acc_ref = ref.get_accumulators()
  e_ref = ref.apply_adam(_hip.VMC_MODE_ENERGY_GRADIENT, 1e-2)

  def rank_fn(rank, coll):
    eng = _engine(shape, theta, cfg, bonds, jx, jz, offset=rank * lb, chains=lb, seed=11)
    eng.set_shift(-40.0)
    eng.epoch_energy_gradient_dist(coll, 2 * n, 3, n, 1e10)
    out = dict(acc=eng.get_accumulators(), grad=eng.get_gradient(_hip.VMC_MODE_ENERGY_GRADIENT),
               shift=eng.get_shift(), configs=eng.get_configs())
    out['energy'] = eng.apply_adam(_hip.VMC_MODE_ENERGY_GRADIENT, 1e-2)
    out['theta'] = eng.get_params()
    eng.close()
    return out

  res, rv = run_ranks(WORLD, rank_fn)
  assert rv.calls == [2] * WORLD                      # one MAX (update_norm) + one SUM (accumulators)
  p = theta.size
  for r, o in enumerate(res):
    np.testing.assert_array_equal(o['configs'], ref.get_configs()[r * lb:(r + 1) * lb])
    np.testing.assert_array_equal(o['acc'], res[0]['acc'])         # every rank holds the same sums
    np.testing.assert_array_equal(o['theta'], res[0]['theta'])     # ... and takes the identical step
    assert o['shift'] == ref.get_shift() > -40.0                   # MAX over all ranks' chains is exact
    assert o['energy'] == res[0]['energy']
  acc = res[0]['acc']
  assert acc[2 * p + 4] == 3 and acc[2 * p + 1] == 3 * b == acc_ref[2 * p + 1]
  assert abs(res[0]['energy'] - e_ref) < 2e-5 * max(1.0, abs(e_ref))
  assert np.abs(res[0]['grad'] - g_ref).max() < 1e-4 * np.abs(g_ref).max()
  w = _well(g_ref)
  assert w.sum() > 0.7 * w.size and np.abs(res[0]['theta'] - ref.get_params())[w].max() < 5e-5
  ref.close()


def test_eight_rank_threads_log_overlap_epoch():
  """training.py:750-763 on 8 ranks: one in-stream accumulator all-reduce per batch, Adam inside."""
  shape, theta, cfg, bonds, jx, jz = _small()
  n, h, L, b = shape
  lb = b // WORLD
  args = (0.12, 2 * n, 3, n, 1e10, 1e-2, 0.9, 0.99, 1e-8)
  ref = _engine(shape, theta, cfg, bonds, jx, jz, seed=11)
  e_ref = ref.epoch_log_overlap(*args)

  def rank_fn(rank, coll):
    eng = _engine(shape, theta, cfg, bonds, jx, jz, offset=rank * lb, chains=lb, seed=11)
    e = eng.epoch_log_overlap_dist(coll, *args)
    out = dict(energy=e, theta=eng.get_params(), omega=eng.get_params(1), configs=eng.get_configs(),
               adam_t=eng.get_adam_state()[2])
    eng.close()
    return out

  res, rv = run_ranks(WORLD, rank_fn)
  assert rv.calls == [1 + 3] * WORLD                  # update_norm + one per batch
  for r, o in enumerate(res):
    np.testing.assert_array_equal(o['theta'], res[0]['theta'])
    np.testing.assert_array_equal(o['omega'], theta)             # supervisor = the epoch's starting point
    np.testing.assert_array_equal(o['configs'], ref.get_configs()[r * lb:(r + 1) * lb])
    assert o['energy'] == res[0]['energy'] and o['adam_t'] == 3
  assert abs(res[0]['energy'] - e_ref) < 2e-5 * max(1.0, abs(e_ref))
  d = np.abs(res[0]['theta'] - ref.get_params())
  assert np.median(d) < 2e-6 and (d < 5e-5).mean() > 0.7      # ill-conditioned Adam entries aside
  ref.close()


def test_eight_rank_threads_stochastic_reconfiguration():
  """SR extension on 8 ranks: the stored samples are sharded, one P+1-float all-reduce per CG iteration."""
  shape, theta, cfg, bonds, jx, jz = _small()
  n, h, L, b = shape
  lb = b // WORLD
  ref = _engine(shape, theta, cfg, bonds, jx, jz, seed=11)
  ref.sr_reserve(2)
  ref.epoch_energy_gradient(n, 2, n, 0.0)
  it_ref, res_ref = ref.sr_solve(0.01, 1e-4, 60)
  x_ref = ref.sr_get_solution()

  def rank_fn(rank, coll):
    eng = _engine(shape, theta, cfg, bonds, jx, jz, offset=rank * lb, chains=lb, seed=11)
    eng.sr_reserve(2)
    eng.epoch_energy_gradient_dist(coll, n, 2, n, 0.0)
    it, rel = eng.sr_solve_dist(coll, 0.01, 1e-4, 60)
    out = dict(it=it, rel=rel, x=eng.sr_get_solution())
    eng.sr_apply(0.05)
    out['theta'] = eng.get_params()
    eng.close()
    return out

  res, rv = run_ranks(WORLD, rank_fn)
  assert rv.calls == [1 + res[0]['it']] * WORLD       # the accumulators, then one per CG iteration
  for o in res:
    assert o['it'] == res[0]['it'] and o['rel'] == res[0]['rel']
    np.testing.assert_array_equal(o['x'], res[0]['x'])
    np.testing.assert_array_equal(o['theta'], res[0]['theta'])
  assert abs(res[0]['it'] - it_ref) <= 1 and res[0]['rel'] <= 1e-4 * 1.01 and res_ref <= 1e-4 * 1.01
  assert np.abs(res[0]['x'] - x_ref).max() < 2e-3 * np.abs(x_ref).max()
  ref.close()


def test_eight_rank_threads_evaluation():
  """evaluation.py:113-152 on 8 ranks through vmc_evaluate: the batch means over ALL ranks' chains."""
  shape, theta, cfg, bonds, jx, jz = _small()
  n, h, L, b = shape
  lb = b // WORLD
  ref = _engine(shape, theta, cfg, bonds, jx, jz, seed=11)
  m_ref, acc_ref = ref.evaluate(None, 3 * n, 7, n)

  def rank_fn(rank, coll):
    eng = _engine(shape, theta, cfg, bonds, jx, jz, offset=rank * lb, chains=lb, seed=11)
    m, a = eng.evaluate(coll, 3 * n, 7, n)
    out = dict(means=m, accepted=a, configs=eng.get_configs())
    eng.close()
    return out

  res, rv = run_ranks(WORLD, rank_fn)
  assert rv.calls == [1] * WORLD                      # ONE float64 all-reduce for the whole evaluation
  for r, o in enumerate(res):
    np.testing.assert_array_equal(o['means'], res[0]['means'])
    np.testing.assert_array_equal(o['configs'], ref.get_configs()[r * lb:(r + 1) * lb])
  np.testing.assert_allclose(res[0]['means'], m_ref, rtol=1e-12, atol=1e-12)   # float64 sums of the same float32 E_loc
  assert sum(o['accepted'] for o in res) == acc_ref
  ref.close()


@pytest.mark.parametrize('name', sorted(JOBS))
def test_eight_rank_threads_full_size_epoch(name):
  """Config 4 / config 5 as 8 concurrent ranks: one EnergyGradient epoch slice (equilibration sweep,
  update_norm, 2 x [accumulate, sweep], all-reduce) + Adam against the engine holding the whole batch."""
  from cgs_vmc_amd import _hip
  shape, theta, cfg, bonds, jx, jz = _inputs(name)
  n, h, L, b = shape
  lb = b // WORLD
  ref = _engine(shape, theta, cfg, bonds, jx, jz)
  ref.epoch_energy_gradient(n, 2, n, 1e10)
  g_ref = ref.get_gradient(_hip.VMC_MODE_ENERGY_GRADIENT)
  acc_ref = ref.get_accumulators().astype(np.float64)
  e_ref = ref.apply_adam(_hip.VMC_MODE_ENERGY_GRADIENT, 1e-3)
  cfg_ref, shift_ref, theta_ref = ref.get_configs(), ref.get_shift(), ref.get_params()
  ref.close()

  def rank_fn(rank, coll):
    eng = _engine(shape, theta, cfg, bonds, jx, jz, offset=rank * lb, chains=lb)
    eng.epoch_energy_gradient_dist(coll, n, 2, n, 1e10)
    out = dict(grad=eng.get_gradient(_hip.VMC_MODE_ENERGY_GRADIENT), shift=eng.get_shift(),
               configs=eng.get_configs(), acc=eng.get_accumulators())
    out['energy'] = eng.apply_adam(_hip.VMC_MODE_ENERGY_GRADIENT, 1e-3)
    out['theta'] = eng.get_params()
    eng.close()
    return out

  res, rv = run_ranks(WORLD, rank_fn)
  assert rv.calls == [2] * WORLD
  for r, o in enumerate(res):
    np.testing.assert_array_equal(o['configs'], cfg_ref[r * lb:(r + 1) * lb], err_msg='rank %d' % r)
    np.testing.assert_array_equal(o['theta'], res[0]['theta'])
    assert o['shift'] == shift_ref and o['energy'] == res[0]['energy']
  assert abs(res[0]['energy'] - e_ref) < 2e-5 * max(1.0, abs(e_ref))
  # the sums over 2 x B samples, in fp32, in two different orders (8 shards of B/8 against one engine
  # of B): the difference is relative to the SUMS; the gradient g2 - <E> g1 is a difference of them
  p = theta.size
  acc = res[0]['acc'].astype(np.float64)
  scale = np.abs(acc_ref[:2 * p]).max()
  assert acc[2 * p + 4] == 2 and acc[2 * p + 1] == 2 * b == acc_ref[2 * p + 1]
  assert np.abs(acc[:2 * p] - acc_ref[:2 * p]).max() < 1e-4 * scale
  e_mean = abs(acc_ref[2 * p] / acc_ref[2 * p + 1])
  g_err = np.abs(res[0]['grad'] - g_ref).max()
  assert g_err < 1e-4 * scale * (1.0 + e_mean) / 2, (g_err, scale, e_mean)
  # Adam's first step is lr g / (|g| + eps'): where |g| is well above that error both took the same step
  w = np.abs(g_ref) > 20 * g_err
  assert w.sum() >= 10, w.sum()      # (the deep config-5 network at random init: a handful of large components)
  assert np.abs(res[0]['theta'] - theta_ref)[w].max() < 5e-5
