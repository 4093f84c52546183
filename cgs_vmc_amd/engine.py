"""VmcEngine: numpy-facing wrapper over one vmc_ctx (one GPU) of libcgsvmc_hip.so.

This is the only module that talks to the C ABI; wavefunctions / operators /
graph_builders / training / evaluation dispatch their op handles to it.
"""
from __future__ import annotations

import atexit
import ctypes as C
import sys
import weakref
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _hip


def _fptr(a: Optional[np.ndarray]):
  if a is None:
    return None
  assert a.dtype == np.float32 and a.flags['C_CONTIGUOUS']
  return a.ctypes.data_as(C.POINTER(C.c_float))


def _iptr(a: np.ndarray):
  assert a.dtype == np.int32 and a.flags['C_CONTIGUOUS']
  return a.ctypes.data_as(C.POINTER(C.c_int32))


_LIVE_ENGINES = weakref.WeakSet()


def _close_live_engines():
  par = sys.modules.get(__package__ + '.parallel')
  if par is not None:                # the library's RCCL communicator goes first (ncclCommDestroy)
    par.close_collective()
  for eng in list(_LIVE_ENGINES):
    try:
      eng.close()
    except Exception:  # pylint: disable=broad-except
      pass


atexit.register(_close_live_engines)


class VmcEngine:
  """Owns the device state of one shard of Markov chains and one ansatz."""

  def __init__(self, n_sites: int, batch_size: int, num_layers: int, layer_size: int,
               nonlinearity: str = 'relu', output_activation: str = 'exp', device: int = 0,
               chain_offset: int = 0, seed: int = 2024, stream: int = 0,
               ansatz: str = 'fully_connected', kernel_size: int = 0, size_x: int = 0,
               size_y: int = 0):
    """Dense ansatz types: num_layers / layer_size = num_fc_layers / fc_layer_size.  Convolutional
    ones ('conv_2d', 'res_net_2d'): num_layers = num_conv_layers or num_resnet_blocks, layer_size =
    num_conv_filters, plus kernel_size and the lattice size_x x size_y (= n_sites)."""
    self._lib = _hip.load()
    self._ctx = C.c_void_p()
    for name, act in (('nonlinearity', nonlinearity), ('output_activation', output_activation)):
      if act not in _hip.ACT_IDS:
        raise ValueError('unknown {} {!r}'.format(name, act))
    if ansatz not in _hip.ANSATZ_IDS:
      raise NotImplementedError('ansatz {!r} has no HIP kernels'.format(ansatz))
    desc = _hip.VmcDesc(n_sites, batch_size, num_layers, layer_size,
                        _hip.ACT_IDS[nonlinearity], _hip.ACT_IDS[output_activation], device,
                        chain_offset, _hip.ANSATZ_IDS[ansatz], 0, seed, stream or None,
                        kernel_size, size_x, size_y, 0)
    rc = self._lib.vmc_create(C.byref(desc), C.byref(self._ctx))
    if rc != _hip.VMC_OK:
      msg = self._lib.vmc_last_error(None).decode()
      self._ctx = C.c_void_p()
      self._raise(rc, msg)
    _LIVE_ENGINES.add(self)
    self.n_sites, self.batch_size = n_sites, batch_size
    self.num_layers, self.layer_size = num_layers, layer_size
    self.chain_offset, self.seed, self.device = chain_offset, seed, device
    self.stream = int(stream or 0)     # hipStream_t the ctx launches on (0: the null stream)
    self.ansatz = ansatz
    self.kernel_size, self.size_x, self.size_y = kernel_size, size_x, size_y
    if ansatz in _hip.CONV_ANSATZ:
      self.num_params = int(self._lib.vmc_num_params_conv(_hip.ANSATZ_IDS[ansatz], num_layers,
                                                          layer_size, kernel_size))
    else:
      self.num_params = int(self._lib.vmc_num_params_ansatz(_hip.ANSATZ_IDS[ansatz], n_sites,
                                                            layer_size, num_layers))
    self.n_bonds = 0

  # ------------------------------------------------------------------ plumbing
  @staticmethod
  def _raise(rc, msg):
    if rc == _hip.VMC_ERR_INVALID:
      raise ValueError(msg)
    if rc == _hip.VMC_ERR_UNSUPPORTED:
      raise NotImplementedError(msg)
    raise _hip.HipLibraryError('libcgsvmc_hip error {}: {}'.format(rc, msg))

  def _check(self, rc):
    if rc != _hip.VMC_OK:
      self._raise(rc, self._lib.vmc_last_error(self._ctx).decode())

  def close(self):
    if getattr(self, '_ctx', None) is not None and self._ctx.value:
      coll = getattr(self, '_coll', None)
      if coll is not None:           # zero-copy views of this ctx's device buffers die with it
        coll.drop_views()
        self._coll = None
      self._lib.vmc_destroy(self._ctx)
      self._ctx = C.c_void_p()

  def __del__(self):
    # engines still alive when the interpreter shuts down are closed by _close_live_engines (atexit,
    # i.e. while the HIP runtime and librccl are certainly still up); a finalizer that runs later
    # than that must not call into them any more
    if sys.is_finalizing():
      return
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass

  # ------------------------------------------------------------------ state
  def set_bonds(self, bonds: Sequence[Tuple[int, int]], j_x, j_z):
    ij = np.ascontiguousarray(np.asarray(bonds, dtype=np.int32).reshape(-1, 2))
    nb = ij.shape[0]
    jx = np.ascontiguousarray(np.broadcast_to(np.asarray(j_x, np.float32), (nb,)))
    jz = np.ascontiguousarray(np.broadcast_to(np.asarray(j_z, np.float32), (nb,)))
    self._check(self._lib.vmc_set_bonds(self._ctx, nb, _iptr(ij), _fptr(jx), _fptr(jz)))
    self.n_bonds = nb

  def set_params(self, theta: np.ndarray, which: int = _hip.VMC_PSI):
    theta = np.ascontiguousarray(theta, dtype=np.float32).ravel()
    if theta.size != self.num_params:
      raise ValueError('expected {} parameters, got {}'.format(self.num_params, theta.size))
    self._check(self._lib.vmc_set_params(self._ctx, which, _fptr(theta)))

  def get_params(self, which: int = _hip.VMC_PSI) -> np.ndarray:
    theta = np.empty(self.num_params, np.float32)
    self._check(self._lib.vmc_get_params(self._ctx, which, _fptr(theta)))
    return theta

  def transfer_params(self):
    self._check(self._lib.vmc_transfer_params(self._ctx))

  def set_configs(self, configs: np.ndarray):
    configs = np.ascontiguousarray(configs, dtype=np.float32)
    if configs.shape != (self.batch_size, self.n_sites):
      raise ValueError('Size of existing variable does not match.')
    self._check(self._lib.vmc_set_configs(self._ctx, _fptr(configs)))

  def get_configs(self) -> np.ndarray:
    out = np.empty((self.batch_size, self.n_sites), np.float32)
    self._check(self._lib.vmc_get_configs(self._ctx, _fptr(out)))
    return out

  def set_shift(self, shift: float, which: int = _hip.VMC_PSI):
    self._check(self._lib.vmc_set_shift(self._ctx, which, float(shift)))

  def get_shift(self, which: int = _hip.VMC_PSI) -> float:
    v = C.c_float()
    self._check(self._lib.vmc_get_shift(self._ctx, which, C.byref(v)))
    return float(v.value)

  # ------------------------------------------------------------------ hot path
  def amplitude(self, configs: Optional[np.ndarray] = None, which: int = _hip.VMC_PSI):
    """Returns (logit, psi) on `configs` [M,N] or on the engine's chains."""
    if configs is None:
      n = self.batch_size
      cp = None
    else:
      configs = np.ascontiguousarray(configs, dtype=np.float32)
      if configs.ndim != 2 or configs.shape[1] != self.n_sites:
        raise ValueError('Input tensor has wrong shape.')
      n = configs.shape[0]
      cp = _fptr(configs)
    logit = np.empty(n, np.float32)
    psi = np.empty(n, np.float32)
    self._check(self._lib.vmc_amplitude(self._ctx, which, cp, n, _fptr(logit), _fptr(psi)))
    return logit, psi

  def mc_steps(self, n_steps: int, want_accepted: bool = True) -> int:
    acc = C.c_int64(0)
    self._check(self._lib.vmc_mc_steps(self._ctx, int(n_steps),
                                       C.byref(acc) if want_accepted else None))
    return int(acc.value)

  def mc_step_injected(self, i_up, i_dn, u) -> np.ndarray:
    i_up = np.ascontiguousarray(i_up, np.int32)
    i_dn = np.ascontiguousarray(i_dn, np.int32)
    u = np.ascontiguousarray(u, np.float32)
    mask = np.empty(self.batch_size, np.uint8)
    self._check(self._lib.vmc_mc_step_injected(
        self._ctx, _iptr(i_up), _iptr(i_dn), _fptr(u), mask.ctypes.data_as(C.POINTER(C.c_uint8))))
    return mask.astype(bool)

  def debug_proposals(self, step: int):
    i_up = np.empty(self.batch_size, np.int32)
    i_dn = np.empty(self.batch_size, np.int32)
    u = np.empty(self.batch_size, np.float32)
    self._check(self._lib.vmc_debug_proposals(self._ctx, int(step), _iptr(i_up), _iptr(i_dn),
                                              _fptr(u)))
    return i_up, i_dn, u

  def debug_sweep_profile(self, n_steps: int):
    """Mean shader cycles per mc_step of the sweep kernel's phases (diagnostic build)."""
    out = (C.c_double * 16)()
    self._check(self._lib.vmc_debug_sweep_profile(self._ctx, int(n_steps), out))
    # 3 'hidden_tail' = what of the hidden layers is not covered by the sub-phases 7..14
    names = ('proposals', 'barrier0', 'build_z1', 'hidden_tail', 'output_dot', 'accept',
             'barrier1', 'l0_ring_prologue', 'l0_barrier', 'l0_resident_mfma',
             'l0_streamed_mfma', 'l0_epilogue', 'l1_barrier', 'l1_mfma', 'l1_epilogue', 'unused')
    if self.sweep_tile() == 8:      # k_sweep8's phases (csrc/sweep8.hip)
      names = ('resolve', 'proposals', 'barrier_a', 'build_draw', 'l0_barrier', 'l0_mfma', 'l0_epilogue',
               'lx_barrier', 'lx_mfma', 'lx_epilogue', 'end_barrier', 'u11', 'u12', 'u13', 'u14', 'u15')
    return dict(zip(names, [float(x) for x in out]))

  @property
  def step_counter(self) -> int:
    v = C.c_uint64()
    self._check(self._lib.vmc_get_step_counter(self._ctx, C.byref(v)))
    return int(v.value)

  @step_counter.setter
  def step_counter(self, step: int):
    self._check(self._lib.vmc_set_step_counter(self._ctx, int(step)))

  def local_energy(self, which: int = _hip.VMC_PSI, want_eloc: bool = True):
    """Returns (eloc[B] or None, mean)."""
    eloc = np.empty(self.batch_size, np.float32) if want_eloc else None
    mean = C.c_double()
    self._check(self._lib.vmc_local_energy(self._ctx, which, _fptr(eloc), C.byref(mean)))
    return eloc, float(mean.value)

  def local_energy_terms(self, which: int = _hip.VMC_PSI):
    diag = np.empty(self.batch_size, np.float32)
    off = np.empty(self.batch_size, np.float32)
    self._check(self._lib.vmc_local_energy_terms(self._ctx, which, _fptr(diag), _fptr(off)))
    return diag, off

  def last_connected_rows(self) -> int:
    v = C.c_int64()
    self._check(self._lib.vmc_last_connected_rows(self._ctx, C.byref(v)))
    return int(v.value)

  def accumulate(self, mode: int, beta: float = 0.0):
    self._check(self._lib.vmc_accumulate(self._ctx, mode, float(beta)))

  def reset_accumulators(self):
    self._check(self._lib.vmc_reset_accumulators(self._ctx))

  def accumulators_devptr(self) -> Tuple[int, int]:
    p = C.c_void_p()
    n = C.c_int64()
    self._check(self._lib.vmc_accumulators_devptr(self._ctx, C.byref(p), C.byref(n)))
    return int(p.value), int(n.value)

  def allreduce_accumulators_rccl(self, nccl_comm: int, world_size: int):
    """SUM all-reduce of the accumulator buffer over an existing RCCL communicator handle (the
    C-ABI path for hosts without torch.distributed; parallel.py uses torch's process group)."""
    self._check(self._lib.vmc_allreduce_accumulators(self._ctx, C.c_void_p(nccl_comm), int(world_size)))

  def get_accumulators(self) -> np.ndarray:
    out = np.empty(2 * self.num_params + 8, np.float32)
    self._check(self._lib.vmc_get_accumulators(self._ctx, _fptr(out)))
    return out

  def set_accumulators(self, acc: np.ndarray):
    acc = np.ascontiguousarray(acc, np.float32)
    assert acc.size == 2 * self.num_params + 8
    self._check(self._lib.vmc_set_accumulators(self._ctx, _fptr(acc)))

  def apply_adam(self, mode: int, lr: float, beta1: float = 0.9, beta2: float = 0.99,
                 eps: float = 1e-8) -> float:
    e = C.c_double()
    self._check(self._lib.vmc_apply_adam(self._ctx, mode, lr, beta1, beta2, eps, C.byref(e)))
    return float(e.value)

  def get_gradient(self, mode: int) -> np.ndarray:
    g = np.empty(self.num_params, np.float32)
    self._check(self._lib.vmc_get_gradient(self._ctx, mode, _fptr(g)))
    return g

  def mean_energy(self) -> float:
    e = C.c_double()
    self._check(self._lib.vmc_mean_energy(self._ctx, C.byref(e)))
    return float(e.value)

  def get_adam_state(self):
    m = np.empty(self.num_params, np.float32)
    v = np.empty(self.num_params, np.float32)
    t = C.c_int64()
    self._check(self._lib.vmc_get_adam_state(self._ctx, _fptr(m), _fptr(v), C.byref(t)))
    return m, v, int(t.value)

  def set_adam_state(self, m, v, t):
    m = np.ascontiguousarray(m, np.float32)
    v = np.ascontiguousarray(v, np.float32)
    self._check(self._lib.vmc_set_adam_state(self._ctx, _fptr(m), _fptr(v), int(t)))

  def epoch_energy_gradient(self, n_eq_steps: int, n_batches: int, n_mc_steps: int,
                            max_value: float = 1e10):
    """training.py:608-617 in one host call (everything before apply_gradients)."""
    self._check(self._lib.vmc_epoch_energy_gradient(self._ctx, int(n_eq_steps), int(n_batches),
                                                    int(n_mc_steps), float(max_value)))

  def epoch_log_overlap(self, beta: float, n_eq_steps: int, n_batches: int, n_mc_steps: int,
                        max_value: float, lr: float, beta1: float, beta2: float,
                        eps: float) -> float:
    """training.py:750-763 in one host call; returns the energy of the last batch."""
    e = C.c_double()
    self._check(self._lib.vmc_epoch_log_overlap(self._ctx, float(beta), int(n_eq_steps),
                                                int(n_batches), int(n_mc_steps), float(max_value),
                                                lr, beta1, beta2, eps, C.byref(e)))
    return float(e.value)

  def update_norm(self, max_value: float = 1e10):
    self._check(self._lib.vmc_update_norm(self._ctx, float(max_value)))

  # ------------------------------------------------------------------ chains sharded over ranks
  # `coll` is a parallel.Collective: (RCCL communicator handle or None, world size, host hook).
  def _bind_collective(self, coll):
    """Registers coll's device / host all-reduce hooks (the transports without a communicator of
    the library's own) and returns (comm, world)."""
    coll.raise_if_broken()
    self._coll = coll
    hook = coll.host_hook()
    if getattr(self, '_host_hook', None) is not hook:
      self._check(self._lib.vmc_set_host_allreduce(
          self._ctx, hook if hook is not None else _hip.HOST_ALLREDUCE_FN(), None))
      # this binding's host hook reduces float64 buffers too (op VMC_REDUCE_SUM_F64: vmc_evaluate's means);
      # a hook that has not said so gets VMC_ERR_UNSUPPORTED instead of doubles it would read as floats
      self._check(self._lib.vmc_set_host_allreduce_caps(
          self._ctx, _hip.VMC_HOST_REDUCE_CAP_F64 if hook is not None else 0))
      self._host_hook = hook       # keeps the ctypes thunk alive as long as the ctx may call it
    dhook = coll.device_hook()
    if getattr(self, '_device_hook', None) is not dhook:
      self._check(self._lib.vmc_set_device_allreduce(
          self._ctx, dhook if dhook is not None else _hip.DEVICE_ALLREDUCE_FN(), None))
      self._device_hook = dhook
    return C.c_void_p(coll.comm or None), int(coll.world)

  def evaluate(self, coll, n_eq_steps: int, n_samples: int, n_mc_steps: int):
    """evaluation.py:113-152 in one host call -> (batch means over ALL ranks' chains [n_samples]
    float64, acceptance count of this rank's chains over the measurement steps).  `coll` is a
    parallel.Collective or None (single rank)."""
    comm, world = self._bind_collective(coll) if coll is not None else (C.c_void_p(None), 1)
    means = np.empty(max(int(n_samples), 1), np.float64)
    acc = C.c_int64(0)
    self._check(self._lib.vmc_evaluate(self._ctx, comm, world, int(n_eq_steps), int(n_samples),
                                       int(n_mc_steps), means.ctypes.data_as(C.POINTER(C.c_double)),
                                       C.byref(acc)))
    return means[:int(n_samples)], int(acc.value)

  def allreduce_accumulators_dist(self, coll):
    comm, world = self._bind_collective(coll)
    self._check(self._lib.vmc_allreduce_accumulators(self._ctx, comm, world))

  def update_norm_dist(self, coll, max_value: float = 1e10):
    comm, world = self._bind_collective(coll)
    self._check(self._lib.vmc_update_norm_dist(self._ctx, comm, world, float(max_value)))

  def debug_allreduce(self, coll, values: np.ndarray, op: str = 'sum') -> np.ndarray:
    comm, world = self._bind_collective(coll)
    buf = np.ascontiguousarray(values, np.float32).copy()
    self._check(self._lib.vmc_debug_allreduce(
        self._ctx, comm, world, _fptr(buf), buf.size,
        _hip.VMC_REDUCE_MAX if op == 'max' else _hip.VMC_REDUCE_SUM))
    return buf

  def epoch_energy_gradient_dist(self, coll, n_eq_steps: int, n_batches: int, n_mc_steps: int,
                                 max_value: float = 1e10):
    """epoch_energy_gradient over sharded chains; the accumulators come back all-reduced."""
    comm, world = self._bind_collective(coll)
    self._check(self._lib.vmc_epoch_energy_gradient_dist(
        self._ctx, comm, world, int(n_eq_steps), int(n_batches), int(n_mc_steps), float(max_value)))

  def epoch_log_overlap_dist(self, coll, beta: float, n_eq_steps: int, n_batches: int,
                             n_mc_steps: int, max_value: float, lr: float, beta1: float,
                             beta2: float, eps: float) -> float:
    """epoch_log_overlap over sharded chains: one in-stream all-reduce per batch."""
    comm, world = self._bind_collective(coll)
    e = C.c_double()
    self._check(self._lib.vmc_epoch_log_overlap_dist(
        self._ctx, comm, world, float(beta), int(n_eq_steps), int(n_batches), int(n_mc_steps),
        float(max_value), lr, beta1, beta2, eps, C.byref(e)))
    return float(e.value)

  def sr_solve_dist(self, coll, diag_shift: float, tol: float, max_iter: int) -> Tuple[int, float]:
    """sr_solve with the stored samples sharded over ranks (one in-stream all-reduce per iteration)."""
    comm, world = self._bind_collective(coll)
    it = C.c_int32()
    res = C.c_double()
    self._check(self._lib.vmc_sr_solve_dist(self._ctx, comm, world, float(diag_shift), float(tol),
                                            int(max_iter), C.byref(it), C.byref(res)))
    return int(it.value), float(res.value)

  # ------------------------------------------------------------------ stochastic reconfiguration
  # Extension named by the north star, absent from the reference (see include/cgsvmc.h).
  def sr_reserve(self, n_batches: int):
    """Sample store for `n_batches` accumulate calls; switches recording on (0 = off)."""
    self._check(self._lib.vmc_sr_reserve(self._ctx, int(n_batches)))

  def sr_num_stored(self) -> int:
    n = C.c_int32()
    self._check(self._lib.vmc_sr_num_stored(self._ctx, C.byref(n)))
    return int(n.value)

  def sr_begin(self) -> float:
    rr = C.c_double()
    self._check(self._lib.vmc_sr_begin(self._ctx, C.byref(rr)))
    return float(rr.value)

  def sr_matvec_partial(self):
    self._check(self._lib.vmc_sr_matvec_partial(self._ctx))

  def sr_matvec_phase1(self):
    """General convolution path: O_b . p of this rank's samples, their sum in the buffer's last float (all-reduce it)."""
    self._check(self._lib.vmc_sr_matvec_phase1(self._ctx))

  def sr_matvec_phase2(self):
    """The matvec proper (weights centred on the all-reduced mean on the general convolution path)."""
    self._check(self._lib.vmc_sr_matvec_phase2(self._ctx))

  def sr_buffer_devptr(self) -> Tuple[int, int]:
    p = C.c_void_p()
    n = C.c_int64()
    self._check(self._lib.vmc_sr_buffer_devptr(self._ctx, C.byref(p), C.byref(n)))
    return int(p.value), int(n.value)

  def sr_get_buffer(self) -> np.ndarray:
    out = np.empty(self.num_params + 1, np.float32)
    self._check(self._lib.vmc_sr_get_buffer(self._ctx, _fptr(out)))
    return out

  def sr_set_buffer(self, buf: np.ndarray):
    buf = np.ascontiguousarray(buf, np.float32)
    assert buf.size == self.num_params + 1
    self._check(self._lib.vmc_sr_set_buffer(self._ctx, _fptr(buf)))

  def sr_cg_update(self, diag_shift: float) -> float:
    rr = C.c_double()
    self._check(self._lib.vmc_sr_cg_update(self._ctx, float(diag_shift), C.byref(rr)))
    return float(rr.value)

  def sr_solve(self, diag_shift: float, tol: float, max_iter: int) -> Tuple[int, float]:
    """(S + diag_shift I) x = f by matrix-free CG on this GPU -> (iterations, |r|/|f|)."""
    it = C.c_int32()
    res = C.c_double()
    self._check(self._lib.vmc_sr_solve(self._ctx, float(diag_shift), float(tol), int(max_iter),
                                       C.byref(it), C.byref(res)))
    return int(it.value), float(res.value)

  def sr_get_solution(self) -> np.ndarray:
    x = np.empty(self.num_params, np.float32)
    self._check(self._lib.vmc_sr_get_solution(self._ctx, _fptr(x)))
    return x

  def sr_apply(self, lr: float) -> float:
    e = C.c_double()
    self._check(self._lib.vmc_sr_apply(self._ctx, float(lr), C.byref(e)))
    return float(e.value)

  def sr_debug_matvec(self, v: np.ndarray, diag_shift: float = 0.0) -> np.ndarray:
    v = np.ascontiguousarray(v, np.float32)
    assert v.size == self.num_params
    out = np.empty(self.num_params, np.float32)
    self._check(self._lib.vmc_sr_debug_matvec(self._ctx, _fptr(v), float(diag_shift), _fptr(out)))
    return out

  # ------------------------------------------------------------------ timing / debug
  def timing_enable(self, on=True):
    """True / 1: every region; 2: only the two roofline kernels (sweep, tail_eloc); False: off."""
    self._check(self._lib.vmc_timing_enable(self._ctx, int(on)))

  def timing_reset(self):
    self._check(self._lib.vmc_timing_reset(self._ctx))

  def timing_get(self, name: str) -> Tuple[float, int]:
    ms = C.c_double()
    n = C.c_int64()
    self._check(self._lib.vmc_timing_get(self._ctx, name.encode(), C.byref(ms), C.byref(n)))
    return float(ms.value), int(n.value)

  def kernel_path(self) -> int:
    """0 fused (<= 256 units), 1 fused with LDS operands (257..512), 2 general path, 3 conv."""
    v = C.c_int32()
    self._check(self._lib.vmc_debug_kernel_path(self._ctx, C.byref(v)))
    return int(v.value)

  def sweep_tile(self, set_to: int = 0) -> int:
    """Chains per sampler workgroup (16: k_sweep16, 8: k_sweep8); set_to = 8 / 16 switches (test hook)."""
    v = C.c_int32()
    self._check(self._lib.vmc_debug_sweep_tile(self._ctx, int(set_to), C.byref(v)))
    return int(v.value)

  def conv_patch(self, n_steps: int) -> bool:
    """Would mc_steps(n_steps) run the patch sampler of the general convolution path (csrc/conv_patch.hip)?"""
    v = C.c_int32()
    self._check(self._lib.vmc_debug_conv_patch(self._ctx, int(n_steps), C.byref(v)))
    return bool(v.value)

  def synchronize(self):
    self._check(self._lib.vmc_synchronize(self._ctx))

  def debug_gemm(self, a: np.ndarray, b: np.ndarray, trans_a=False, trans_b=False) -> np.ndarray:
    """C = op(A) op(B) through the library's MFMA GEMM (test hook)."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    if trans_a:
      k, m = a.shape; sam, sak = 1, m
    else:
      m, k = a.shape; sam, sak = k, 1
    if trans_b:
      n, k2 = b.shape; sbk, sbn = 1, k2
    else:
      k2, n = b.shape; sbk, sbn = n, 1
    assert k == k2
    c = np.empty((m, n), np.float32)
    self._check(self._lib.vmc_debug_gemm(self._ctx, m, n, k, _fptr(a), sam, sak, a.size, _fptr(b),
                                         sbk, sbn, b.size, _fptr(c)))
    return c
