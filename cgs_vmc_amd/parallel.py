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
import sys

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

  def __init__(self, ptr: int, n: int, typestr: str = '<f4'):
    self.__cuda_array_interface__ = {
        'shape': (n,), 'typestr': typestr, 'data': (ptr, False), 'version': 2, 'strides': None}


def _torch_stream(device: int, stream: int):
  """torch's handle of the hipStream_t a ctx launches on (vmc_desc.stream; 0 = the null stream, which
  is torch's default stream).  Collectives issued under `with torch.cuda.stream(...)` of it are
  ordered against the library's kernels without a host synchronisation."""
  import torch
  if stream:
    return torch.cuda.ExternalStream(int(stream), device=torch.device('cuda', device))
  return torch.cuda.default_stream(torch.device('cuda', device))


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
  # Stream ordering, no host sync: the collective is issued with the ctx's own stream
  # (vmc_desc.stream; the null stream = torch's default stream unless the caller chose one) as
  # torch's current stream: RCCL waits for that stream before it starts and makes it wait for
  # the result (torch.distributed semantics for async_op=False).
  # CGS_VMC_SAFE_SYNC=1 adds explicit host synchronisation on both sides.
  import torch
  safe = os.environ.get('CGS_VMC_SAFE_SYNC', '0') == '1'
  if safe:
    engine.synchronize()
  t = accumulator_tensor(engine)
  st = _torch_stream(engine.device, getattr(engine, 'stream', 0))
  with torch.cuda.stream(st):
    _dist().all_reduce(t, op=_dist().ReduceOp.SUM)
    # mean_tensor's count is the number of accumulate CALLS (training.py:550-553), which every
    # rank made in lock-step: undo the sum so that sharded == unsharded gradients
    t[t.numel() - 4] /= world_size()
  if safe:
    st.synchronize()


class _PendingReduce:
  """Handle of an in-flight accumulator all-reduce (see allreduce_accumulators_begin)."""

  def __init__(self, work, tensor, stream=None):
    self.work, self.tensor, self.stream = work, tensor, stream

  def wait(self):
    """Orders the ctx's stream after the collective (no host sync) and restores g_count."""
    if self.work is not None:
      import torch
      with torch.cuda.stream(self.stream):
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
  import torch
  t = accumulator_tensor(engine)
  st = _torch_stream(engine.device, getattr(engine, 'stream', 0))
  with torch.cuda.stream(st):
    work = _dist().all_reduce(t, op=_dist().ReduceOp.SUM, async_op=True)
  return _PendingReduce(work, t, st)


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
  with torch.cuda.stream(_torch_stream(engine.device, getattr(engine, 'stream', 0))):
    _dist().all_reduce(t, op=_dist().ReduceOp.SUM)


class Collective:
  """How the LIBRARY reduces over ranks inside its device-resident epoch / CG / evaluation entry
  points (include/cgsvmc.h, `*_dist`, vmc_evaluate).  Three transports:

  * 'torch' (default under backend 'nccl'): a DEVICE hook -- the library calls back at the point of
    the epoch where the all-reduce belongs and the hook issues torch.distributed.all_reduce (RCCL
    over xGMI, torch's own communicator) on a zero-copy view of the device buffer with the ctx's
    stream as torch's current stream: in stream, no staging, no host synchronisation;
  * 'rccl' (CGS_VMC_TRANSPORT=rccl): an RCCL communicator the library itself created next to
    torch's (in-stream ncclAllReduce, no Python in the loop).  Opt-in: it has run on one rank only
    (tests/test_gpu_dist.py) -- no multi-GPU box has been available to this repository;
  * 'host' (any other backend, e.g. the gloo tests; CGS_VMC_TRANSPORT=host): a host hook that
    all-reduces a pinned staging buffer through torch.distributed.

  `comm` is the ncclComm_t as an integer (0: none), `world` the number of ranks."""

  def __init__(self, comm: int = 0, world: int = 1, use_hook: bool = False,
               use_device_hook: bool = False, device: int = 0):
    self.comm, self.world, self.device = int(comm), int(world), int(device)
    self._hook = None
    self._dhook = None
    self._views = {}
    self._broken = None          # the first exception raised inside a hook (the C call only sees a code)
    from . import _hip
    if use_hook:
      self._hook = _hip.HOST_ALLREDUCE_FN(self._host_allreduce)
    if use_device_hook:
      self._dhook = _hip.DEVICE_ALLREDUCE_FN(self._device_allreduce)

  @property
  def transport(self) -> str:
    if self.comm:
      return 'rccl'
    if self.world <= 1:
      return 'none'
    return 'torch' if self._dhook is not None else 'host'

  def host_hook(self):
    return self._hook

  def device_hook(self):
    return self._dhook

  def allreduce_host(self, buf: np.ndarray, op: str = 'sum') -> np.ndarray:
    """In-place all-reduce of a float32 / float64 host array over torch.distributed (what the
    host hook does)."""
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
      if op == 2:                                # VMC_REDUCE_SUM_F64: n doubles in the float staging buffer
        import ctypes as C
        buf = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(int(n),))
      else:
        buf = np.ctypeslib.as_array(ptr, shape=(int(n),))
      self.allreduce_host(buf, 'max' if op == 1 else 'sum')
      return 0
    except Exception as e:  # pylint: disable=broad-except
      return self._hook_failed(e)

  def _device_allreduce(self, user, ptr, n, op, stream):   # vmc_device_allreduce_fn
    try:
      import torch
      dist = _dist()
      key = (int(ptr), int(n), int(op))
      t = self._views.get(key)
      if t is None:
        if len(self._views) > 16:
          self._views.clear()
        t = torch.as_tensor(_DevArray(int(ptr), int(n), '<f8' if op == 2 else '<f4'),
                            device=torch.device('cuda', self.device))
        self._views[key] = t
      with torch.cuda.stream(_torch_stream(self.device, stream or 0)):
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)
      return 0
    except Exception as e:  # pylint: disable=broad-except
      return self._hook_failed(e)

  def _hook_failed(self, exc) -> int:
    """A hook runs inside a C call: the exception cannot cross it, the call returns VMC_ERR_HIP and the
    engine raises HipLibraryError on THIS rank while the others may already sit in the collective.  The
    cause is kept (raise_if_broken re-raises it on the next use of this Collective, which is then
    refused for good), and with CGS_VMC_HOOK_FAILURE=abort the rank leaves at once with exit code 70, so
    that a launcher (torch.distributed.run) tears the job down instead of leaving the peers blocked
    until the backend's own timeout."""
    import traceback
    traceback.print_exc()
    if self._broken is None:
      self._broken = exc
    if self.world > 1 and os.environ.get('CGS_VMC_HOOK_FAILURE', 'raise') == 'abort':
      sys.stderr.write('cgs_vmc_amd.parallel: all-reduce hook failed on rank {}; aborting the rank '
                       '(CGS_VMC_HOOK_FAILURE=abort)\n'.format(rank()))
      sys.stderr.flush()
      os._exit(70)
    return 1

  def raise_if_broken(self):
    if self._broken is not None:
      raise RuntimeError('this Collective failed inside an all-reduce hook and the ranks may be out of step; '
                         'restart the job') from self._broken

  def drop_views(self):
    """Forgets the zero-copy torch views of engine buffers (an engine that closes calls this: its
    device pointers may be handed out again to another allocation of the same size)."""
    self._views.clear()

  def close(self):
    self._views.clear()
    if self.comm:
      from . import _hip
      _hip.load().vmc_rccl_comm_destroy(self.comm)
      self.comm = 0


_collective = None


def transport_choice() -> str:
  """'torch' | 'rccl' | 'host' for the current process group (see Collective)."""
  nccl = _dist().get_backend() == 'nccl'
  choice = os.environ.get('CGS_VMC_TRANSPORT')
  if choice is None:
    if not nccl:
      return 'host'
    choice = 'rccl' if os.environ.get('CGS_VMC_LIBRARY_RCCL', '0') == '1' else 'torch'
  if choice not in ('torch', 'rccl', 'host'):
    raise ValueError("CGS_VMC_TRANSPORT must be 'torch', 'rccl' or 'host', not {!r}".format(choice))
  if choice == 'rccl' and not nccl:
    raise ValueError("CGS_VMC_TRANSPORT=rccl needs the 'nccl' backend (one device per rank)")
  return choice       # 'torch' under gloo: gloo reduces device tensors too (the single-GPU tests)


def collective(device: int = None) -> Collective:
  """The process-wide Collective for the current torch.distributed group (created on first use;
  a collective call: every rank must reach it)."""
  global _collective
  if _collective is not None and _collective.world == world_size():
    return _collective
  if _collective is not None:
    _collective.close()
  if world_size() == 1:
    _collective = Collective()
    return _collective
  dev = local_rank() if device is None else device
  choice = transport_choice()
  if choice == 'host':
    _collective = Collective(0, world_size(), use_hook=True, device=dev)
  elif choice == 'torch':
    _collective = Collective(0, world_size(), use_device_hook=True, device=dev)
  else:
    _collective = rccl_collective(dev)
  return _collective


def close_collective():
  """Destroys the process-wide Collective (ncclCommDestroy for the 'rccl' transport).  Runs at
  interpreter exit BEFORE the engines are closed (engine._close_live_engines)."""
  global _collective
  if _collective is not None:
    try:
      _collective.close()
    except Exception:  # pylint: disable=broad-except
      pass
    _collective = None


def rccl_collective(device: int, world: int = None, rank_: int = None) -> Collective:
  """Creates the library's own RCCL communicator: rank 0 draws the ncclUniqueId, it travels over
  the existing process group (or not at all for world == 1), every rank joins on `device`.  A
  failure on ANY rank (the id on rank 0, ncclCommInitRank anywhere) is raised on EVERY rank: the
  id travels together with rank 0's status, and a success flag is all-reduced after the join."""
  import ctypes as C
  from . import _hip
  lib = _hip.load()
  world = world_size() if world is None else world
  rank_ = rank() if rank_ is None else rank_
  uid = (C.c_uint8 * 128)()
  status = ''
  if rank_ == 0 and lib.vmc_rccl_unique_id(uid) != 0:
    status = 'vmc_rccl_unique_id: ' + lib.vmc_rccl_last_error().decode()
  if world > 1:
    box = [(status, bytes(uid))]
    _dist().broadcast_object_list(box, src=0)
    status, raw = box[0]
    uid = (C.c_uint8 * 128).from_buffer_copy(raw)
  if status:
    raise _hip.HipLibraryError(status)       # every rank raises together, nobody is left waiting
  comm = C.c_void_p()
  err = ''
  if lib.vmc_rccl_comm_create(uid, world, rank_, device, C.byref(comm)) != 0:
    err = 'vmc_rccl_comm_create: ' + lib.vmc_rccl_last_error().decode()
  if world > 1:
    failed = allreduce_sum(1.0 if err else 0.0)
    if failed > 0.0:
      if comm.value:
        lib.vmc_rccl_comm_destroy(comm)
      raise _hip.HipLibraryError(err or 'vmc_rccl_comm_create failed on {:.0f} other rank(s)'.format(failed))
  elif err:
    raise _hip.HipLibraryError(err)
  return Collective(comm.value, world, device=device)


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
  two_phase = hasattr(engine, 'sr_matvec_phase1') and engine.kernel_path() == 6
  while it < max_iter and rr0 > 0.0 and rr > tol * tol * rr0:
    if two_phase:      # general convolutions: the weights O_b . p are centred on their mean over ALL ranks
      engine.sr_matvec_phase1()
      allreduce_sr_buffer(engine)     # (zeros but for the last float, sum_b O_b . p)
      engine.sr_matvec_phase2()
    else:
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
