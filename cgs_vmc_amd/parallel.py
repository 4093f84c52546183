"""Chain sharding across GPUs (absent from the reference, which is
single-process).

Markov chains are independent given theta (graph_builders.py:57-88 has no cross-row op), so
rank r owns chains [r*B/G, (r+1)*B/G) and the only exchange is a SUM all-reduce of the
accumulator buffer [g1 | g2 | e_total e_count r_total r_count g_count ...] (2P+8 floats) once
per optimizer step, plus scalar MAX (update_norm) and SUM (acceptance / evaluation mean).
Collectives go through torch.distributed: backend 'nccl' is RCCL over xGMI on ROCm; 'gloo'
is used by the CPU tests.  Every rank then applies the identical Adam step.
"""
from __future__ import annotations

import os

import numpy as np


def _dist():
  import torch.distributed as dist
  return dist


def is_distributed() -> bool:
  try:
    dist = _dist()
    return dist.is_available() and dist.is_initialized()
  except Exception:  # pylint: disable=broad-except
    return False


def world_size() -> int:
  return _dist().get_world_size() if is_distributed() else 1


def rank() -> int:
  return _dist().get_rank() if is_distributed() else 0


def local_rank() -> int:
  """Device ordinal of this rank (LOCAL_RANK, folded onto the visible devices so that the
  2-rank gloo test can share one GPU)."""
  if not is_distributed():
    return 0
  lr = int(os.environ.get('LOCAL_RANK', '0'))
  try:
    import torch
    n = torch.cuda.device_count()
    return lr % n if n > 0 else lr
  except Exception:  # pylint: disable=broad-except
    return lr


def init_from_env(backend: str = 'nccl'):
  """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* when WORLD_SIZE > 1."""
  if int(os.environ.get('WORLD_SIZE', '1')) <= 1 or is_distributed():
    return
  import torch
  dist = _dist()
  backend = os.environ.get('CGS_VMC_DIST_BACKEND', backend)   # 'gloo' for single-GPU tests
  if backend == 'nccl':
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
  dist.init_process_group(backend=backend)


def shard(batch_size: int):
  """(local_batch, chain_offset) of this rank for a global batch."""
  g, r = world_size(), rank()
  if batch_size % g != 0:
    raise ValueError('batch_size %d is not divisible by world size %d' % (batch_size, g))
  local = batch_size // g
  return local, r * local


class _DevArray:
  """__cuda_array_interface__ view of library-owned device memory (zero copy)."""

  def __init__(self, ptr: int, n: int):
    self.__cuda_array_interface__ = {
        'shape': (n,), 'typestr': '<f4', 'data': (ptr, False), 'version': 2, 'strides': None}


_acc_views = {}


def accumulator_tensor(engine):
  """torch view (device memory of the engine) of the accumulator buffer; cached per buffer."""
  import torch
  ptr, n = engine.accumulators_devptr()
  key = (engine.device, ptr, n)
  t = _acc_views.get(key)
  if t is None:
    t = torch.as_tensor(_DevArray(ptr, n), device=torch.device('cuda', engine.device))
    _acc_views.clear()          # one live engine per process is the normal case
    _acc_views[key] = t
  return t


def allreduce_accumulators(engine):
  """In-place SUM all-reduce of the engine's accumulator buffer over all ranks."""
  if world_size() == 1:
    return
  if _dist().get_backend() != 'nccl':
    # gloo (tests, CPU rendezvous): stage through the host
    engine.set_accumulators(reduce_accumulators_host(engine.get_accumulators()))
    return
  # Stream ordering, no host sync: the library launches on the legacy null stream, which is
  # torch's current (default) stream; the RCCL collective waits for that stream before it
  # starts and makes it wait for the result (torch.distributed semantics for async_op=False).
  # CGS_VMC_SAFE_SYNC=1 adds explicit host synchronisation on both sides.
  safe = os.environ.get('CGS_VMC_SAFE_SYNC', '0') == '1'
  if safe:
    engine.synchronize()
  t = accumulator_tensor(engine)
  _dist().all_reduce(t, op=_dist().ReduceOp.SUM)
  # mean_tensor's count is the number of accumulate CALLS (training.py:550-553), which every
  # rank made in lock-step: undo the sum so that sharded == unsharded gradients
  t[t.numel() - 4] /= world_size()
  if safe:
    import torch
    torch.cuda.current_stream(t.device).synchronize()


class _PendingReduce:
  """Handle of an in-flight accumulator all-reduce (see allreduce_accumulators_begin)."""

  def __init__(self, work, tensor):
    self.work, self.tensor = work, tensor

  def wait(self):
    """Orders the current stream after the collective (no host sync) and restores g_count."""
    if self.work is not None:
      self.work.wait()
      self.tensor[self.tensor.numel() - 4] /= world_size()
      self.work = None


def allreduce_accumulators_begin(engine) -> _PendingReduce:
  """Starts the SUM all-reduce of the accumulator buffer on RCCL's stream and returns at
  once, so that work which does not touch the accumulators (the next MC sweep) overlaps with
  it; call .wait() before the accumulators are read or written again."""
  if world_size() == 1:
    return _PendingReduce(None, None)
  if _dist().get_backend() != 'nccl':
    allreduce_accumulators(engine)
    return _PendingReduce(None, None)
  t = accumulator_tensor(engine)
  return _PendingReduce(_dist().all_reduce(t, op=_dist().ReduceOp.SUM, async_op=True), t)


def allreduce_sr_buffer(engine):
  """In-place SUM all-reduce of the SR matrix-vector buffer (P+1 floats, vmc_sr_buffer_devptr)."""
  if world_size() == 1:
    return
  import torch
  if _dist().get_backend() != 'nccl':     # gloo (tests): stage through the host
    h = torch.from_numpy(np.ascontiguousarray(engine.sr_get_buffer(), np.float32))
    _dist().all_reduce(h, op=_dist().ReduceOp.SUM)
    engine.sr_set_buffer(h.numpy())
    return
  ptr, n = engine.sr_buffer_devptr()
  t = torch.as_tensor(_DevArray(ptr, n), device=torch.device('cuda', engine.device))
  _dist().all_reduce(t, op=_dist().ReduceOp.SUM)


class Collective:
  """How the LIBRARY reduces over ranks inside its device-resident epoch / CG entry points
  (include/cgsvmc.h, `*_dist`): an RCCL communicator the library itself created next to
  torch.distributed's process group (backend 'nccl': in-stream ncclAllReduce over xGMI, no host
  synchronisation), or -- any other backend, e.g. the gloo tests -- a host hook that all-reduces
  a pinned staging buffer through torch.distributed.  `comm` is the ncclComm_t as an integer (0:
  none), `world` the number of ranks."""

  def __init__(self, comm: int = 0, world: int = 1, use_hook: bool = False):
    self.comm, self.world = int(comm), int(world)
    self._hook = None
    if use_hook:
      from . import _hip
      self._hook = _hip.HOST_ALLREDUCE_FN(self._host_allreduce)

  def host_hook(self):
    return self._hook

  def allreduce_host(self, buf: np.ndarray, op: str = 'sum') -> np.ndarray:
    """In-place all-reduce of a float32 host array over torch.distributed (what the hook does)."""
    if self.world > 1:
      import torch
      dist = _dist()
      t = torch.from_numpy(buf)
      if dist.get_backend() == 'nccl':     # host data on an RCCL group: stage through the device
        d = t.to(torch.device('cuda', local_rank()))
        dist.all_reduce(d, op=dist.ReduceOp.MAX if op == 'max' else dist.ReduceOp.SUM)
        t.copy_(d)
      else:
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 'max' else dist.ReduceOp.SUM)
    return buf

  def _host_allreduce(self, user, ptr, n, op):   # vmc_host_allreduce_fn
    try:
      buf = np.ctypeslib.as_array(ptr, shape=(int(n),))
      self.allreduce_host(buf, 'max' if op == 1 else 'sum')
      return 0
    except Exception:  # pylint: disable=broad-except
      import traceback
      traceback.print_exc()
      return 1

  def close(self):
    if self.comm:
      from . import _hip
      _hip.load().vmc_rccl_comm_destroy(self.comm)
      self.comm = 0


_collective = None


def collective(device: int = None) -> Collective:
  """The process-wide Collective for the current torch.distributed group (created on first use;
  a collective call: every rank must reach it)."""
  global _collective
  if _collective is not None and _collective.world == world_size():
    return _collective
  if world_size() == 1:
    _collective = Collective()
    return _collective
  dist = _dist()
  use_rccl = dist.get_backend() == 'nccl' and os.environ.get('CGS_VMC_LIBRARY_RCCL', '1') != '0'
  if not use_rccl:
    _collective = Collective(0, world_size(), use_hook=True)
    return _collective
  _collective = rccl_collective(local_rank() if device is None else device)
  return _collective


def rccl_collective(device: int, world: int = None, rank_: int = None) -> Collective:
  """Creates the library's own RCCL communicator: rank 0 draws the ncclUniqueId, it travels over
  the existing process group (or not at all for world == 1), every rank joins on `device`."""
  import ctypes as C
  from . import _hip
  lib = _hip.load()
  world = world_size() if world is None else world
  rank_ = rank() if rank_ is None else rank_
  uid = (C.c_uint8 * 128)()
  if rank_ == 0 and lib.vmc_rccl_unique_id(uid) != 0:
    raise _hip.HipLibraryError('vmc_rccl_unique_id: ' + lib.vmc_rccl_last_error().decode())
  if world > 1:
    box = [bytes(uid)]
    _dist().broadcast_object_list(box, src=0)
    uid = (C.c_uint8 * 128).from_buffer_copy(box[0])
  comm = C.c_void_p()
  if lib.vmc_rccl_comm_create(uid, world, rank_, device, C.byref(comm)) != 0:
    raise _hip.HipLibraryError('vmc_rccl_comm_create: ' + lib.vmc_rccl_last_error().decode())
  return Collective(comm.value, world)


def sr_solve(engine, diag_shift: float, tol: float, max_iter: int):
  """Matrix-free CG for (S + diag_shift I) x = f with the chains (and therefore the stored
  samples) sharded over ranks: one P+1-float SUM all-reduce per iteration; every rank runs the
  identical recurrence on the reduced vectors.  Returns (iterations, |r| / |f|).  The
  accumulators must already be all-reduced (vmc_sr_begin reads f and <O> from them).
  Engines with the device-resident entry (vmc_sr_solve_dist) run the whole loop in one call, the
  all-reduce in stream; the op-by-op loop below remains for engines without it."""
  if world_size() == 1:
    return engine.sr_solve(diag_shift, tol, max_iter)
  if hasattr(engine, 'sr_solve_dist') and os.environ.get('CGS_VMC_DIST_FUSED', '1') != '0':
    return engine.sr_solve_dist(collective(), diag_shift, tol, max_iter)
  rr0 = rr = engine.sr_begin()
  it = 0
  while it < max_iter and rr0 > 0.0 and rr > tol * tol * rr0:
    engine.sr_matvec_partial()
    allreduce_sr_buffer(engine)
    rr = engine.sr_cg_update(diag_shift)
    it += 1
  return it, (float(np.sqrt(rr / rr0)) if rr0 > 0.0 else 0.0)


def allreduce_array(values: np.ndarray, op: str = 'sum') -> np.ndarray:
  """All-reduce of a small host array (float64) with SUM or MAX."""
  values = np.asarray(values, np.float64)
  if world_size() == 1:
    return values
  import torch
  dist = _dist()
  dev = torch.device('cuda', local_rank()) if dist.get_backend() == 'nccl' else torch.device('cpu')
  t = torch.as_tensor(values, dtype=torch.float64, device=dev).clone()
  dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 'sum' else dist.ReduceOp.MAX)
  return t.cpu().numpy()


def allreduce_sum(value: float) -> float:
  return float(allreduce_array(np.array([value]), 'sum')[0])


def allreduce_max(value: float) -> float:
  return float(allreduce_array(np.array([value]), 'max')[0])


def reduce_accumulators_host(acc: np.ndarray) -> np.ndarray:
  """SUM all-reduce of a host copy of an accumulator buffer (any backend; used by the gloo
  tests and as the generic path)."""
  if world_size() == 1:
    return acc
  import torch
  dist = _dist()
  dev = torch.device('cuda', local_rank()) if dist.get_backend() == 'nccl' else torch.device('cpu')
  t = torch.as_tensor(np.ascontiguousarray(acc, np.float32), device=dev).clone()
  dist.all_reduce(t, op=dist.ReduceOp.SUM)
  t[t.numel() - 4] /= world_size()      # g_count: calls, not calls x ranks (see above)
  return t.cpu().numpy()
