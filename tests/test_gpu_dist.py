"""GPU: the device-resident entries for chains sharded over ranks (include/cgsvmc.h, `*_dist`).

(a) with a 1-rank RCCL communicator created by the library itself (vmc_rccl_comm_create) every
    `_dist` entry is BIT-IDENTICAL to its single-rank twin -- the in-stream ncclAllReduce over one
    rank is the identity and `g_count / 1` is exact;
(b) two ranks sharing this GPU over gloo (the host-hook transport; RCCL refuses two ranks on one
    device) run the product routing -- training.run_optimization_epoch / parallel.sr_solve -- and
    match the same epochs of an unsharded engine: chains bit-identical (Philox keyed by global
    chain id), parameters within the fp32 reduction-order tolerance (tests/_dist_gpu_worker.py).
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import vmc_oracle as vo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _engine(seed=11, b=96, sr=0):
  from cgs_vmc_amd.engine import VmcEngine
  n, h, L = 16, 32, 2
  rng = np.random.default_rng(3)
  eng = VmcEngine(n, b, L, h, seed=seed)
  eng.set_params(vo.init_params(n, h, L, rng) + (0.05 * rng.standard_normal(vo.num_params(n, h, L))).astype(np.float32))
  eng.set_configs(vo.random_configurations(n, b, np.random.RandomState(4)))
  eng.set_bonds(vo.torus_bonds(4, 4), -1.0, 1.0)
  if sr:
    eng.sr_reserve(sr)
  return eng


def _state(eng):
  m, v, t = eng.get_adam_state()
  return dict(theta=eng.get_params(), omega_shift=eng.get_shift(1), shift=eng.get_shift(0),
              configs=eng.get_configs(), m=m, v=v, t=t, acc=eng.get_accumulators())


def _assert_same(a, b):
  for k in a:
    np.testing.assert_array_equal(a[k], b[k], err_msg=k)


@pytest.fixture(scope='module')
def rccl1():
  from cgs_vmc_amd import parallel
  coll = parallel.rccl_collective(device=0, world=1, rank_=0)
  assert coll.comm != 0 and coll.world == 1
  yield coll
  if not os.environ.get('CGS_TEST_KEEP_COMM'):
    coll.close()


def test_library_rccl_allreduce_one_rank(rccl1):
  eng = _engine()
  x = np.random.default_rng(0).standard_normal(1000).astype(np.float32)
  np.testing.assert_array_equal(eng.debug_allreduce(rccl1, x, 'sum'), x)
  np.testing.assert_array_equal(eng.debug_allreduce(rccl1, x, 'max'), x)
  eng.close()


def test_log_overlap_epoch_dist_is_bit_identical_with_one_rccl_rank(rccl1):
  """training.py:750-763: [sweep, reset, accumulate, all-reduce, Adam] x n_batches in one call."""
  a, b = _engine(), _engine()
  args = (0.12, 32, 3, 16, 1e10, 1e-2, 0.9, 0.99, 1e-8)
  for _ in range(2):
    e_a = a.epoch_log_overlap(*args)
    e_b = b.epoch_log_overlap_dist(rccl1, *args)
    assert e_a == e_b
    _assert_same(_state(a), _state(b))
  a.close(); b.close()


def test_energy_gradient_epoch_dist_is_bit_identical_with_one_rccl_rank(rccl1):
  from cgs_vmc_amd import _hip
  a, b = _engine(), _engine()
  for _ in range(2):
    a.epoch_energy_gradient(32, 3, 16, 1e10)
    b.epoch_energy_gradient_dist(rccl1, 32, 3, 16, 1e10)
    _assert_same(_state(a), _state(b))
    assert a.apply_adam(_hip.VMC_MODE_ENERGY_GRADIENT, 1e-2) == b.apply_adam(_hip.VMC_MODE_ENERGY_GRADIENT, 1e-2)
    _assert_same(_state(a), _state(b))
  a.close(); b.close()


def test_update_norm_dist_is_bit_identical_with_one_rccl_rank(rccl1):
  a, b = _engine(), _engine()
  for eng in (a, b):
    eng.set_shift(-40.0)          # psi = exp(logit + 40) > 1e10: the shift must move
  a.update_norm(1e10)
  b.update_norm_dist(rccl1, 1e10)
  assert a.get_shift() == b.get_shift() and a.get_shift() > -40.0
  a.close(); b.close()


def test_sr_solve_dist_is_bit_identical_with_one_rccl_rank(rccl1):
  a, b = _engine(sr=2), _engine(sr=2)
  a.epoch_energy_gradient(16, 2, 16, 0.0)
  b.epoch_energy_gradient_dist(rccl1, 16, 2, 16, 0.0)
  ra = a.sr_solve(0.01, 1e-4, 50)
  rb = b.sr_solve_dist(rccl1, 0.01, 1e-4, 50)
  assert ra == rb and ra[0] > 0
  np.testing.assert_array_equal(a.sr_get_solution(), b.sr_get_solution())
  a.sr_apply(0.05); b.sr_apply(0.05)
  np.testing.assert_array_equal(a.get_params(), b.get_params())
  a.close(); b.close()


def test_sharded_world_without_transport_is_refused():
  """world_size > 1 with neither an RCCL communicator nor a host hook must fail loudly."""
  from cgs_vmc_amd import _hip, parallel
  eng = _engine()
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  with pytest.raises(_hip.HipLibraryError, match='vmc_set_host_allreduce'):
    eng.allreduce_accumulators_dist(parallel.Collective(0, 2))
  eng.close()


def test_host_hook_transport_doubles_like_two_identical_ranks():
  """The host-hook path end to end in one process: a hook that behaves like a second rank holding
  the same data (SUM doubles, MAX is the identity)."""
  from cgs_vmc_amd import _hip, parallel

  class Twin(parallel.Collective):
    def allreduce_host(self, buf, op='sum'):
      if op == 'sum':
        buf *= 2.0
      return buf

  coll = Twin(0, 2, use_hook=True)
  eng = _engine()
  x = np.arange(1, 300, dtype=np.float32)
  np.testing.assert_array_equal(eng.debug_allreduce(coll, x, 'sum'), 2 * x)
  np.testing.assert_array_equal(eng.debug_allreduce(coll, x, 'max'), x)
  eng.reset_accumulators()
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  before = eng.get_accumulators()
  eng.allreduce_accumulators_dist(coll)
  expect = 2 * before
  expect[-4] = before[-4]                  # g_count: calls, not calls x ranks
  np.testing.assert_array_equal(eng.get_accumulators(), expect)
  eng.close()


def _free_port():
  s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
  return p


def test_two_gloo_ranks_route_training_epochs_through_the_dist_entries():
  port = _free_port()
  procs = []
  for rank in range(2):
    env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(port), CGS_VMC_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', '_dist_gpu_worker.py')],
                                  env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
  outs = []
  for p in procs:
    try:
      out, _ = p.communicate(timeout=600)
    except subprocess.TimeoutExpired:
      for q in procs:
        q.kill()
      raise
    outs.append(out.decode())
  for rank, (p, out) in enumerate(zip(procs, outs)):
    assert p.returncode == 0 and 'rank {} ok'.format(rank) in out, out[-4000:]
