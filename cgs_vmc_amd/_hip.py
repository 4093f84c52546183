"""ctypes binding of libcgsvmc_hip.so (include/cgsvmc.h).

There is no CPU fallback: if the shared library is missing or no GPU is present the
product path raises.  Build with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C cgs_vmc_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os

_LIB_NAME = 'libcgsvmc_hip.so'
_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)

VMC_OK = 0
VMC_ERR_INVALID = -1
VMC_ERR_UNSUPPORTED = -2
VMC_ERR_HIP = -3
VMC_ERR_STATE = -4

VMC_PSI, VMC_OMEGA = 0, 1
VMC_MODE_ENERGY_GRADIENT, VMC_MODE_LOG_OVERLAP_ITSWO = 0, 1
# wavefunctions.WAVEFUNCTION_TYPES with kernels (cgsvmc.h VMC_ANSATZ_*)
ANSATZ_IDS = {'fully_connected': 0, 'rbm': 1, 'conv_2d': 2, 'res_net_2d': 3, 'conv_1d': 4, 'res_net_1d': 5}
CONV_ANSATZ = ('conv_2d', 'res_net_2d', 'conv_1d', 'res_net_1d')
# layers.NONLINEARITIES ids (cgsvmc.h)
ACT_IDS = {'relu': 0, 'exp': 1, 'cos': 2, 'tan': 3, 'tanh': 4, 'sigmoid': 5, 'identity': 6}


class VmcDesc(C.Structure):
  _fields_ = [
      ('n_sites', C.c_int32), ('batch_size', C.c_int32), ('num_layers', C.c_int32),
      ('layer_size', C.c_int32), ('nonlinearity', C.c_int32),
      ('output_activation', C.c_int32), ('device', C.c_int32), ('chain_offset', C.c_int32),
      ('ansatz', C.c_int32), ('reserved', C.c_int32),
      ('seed', C.c_uint64), ('stream', C.c_void_p),
      ('kernel_size', C.c_int32), ('size_x', C.c_int32), ('size_y', C.c_int32),
      ('reserved2', C.c_int32),
  ]


class HipLibraryError(RuntimeError):
  """The HIP extension is missing or failed; the hot path has no fallback."""


_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
_ctx = C.c_void_p
_bp = C.POINTER(C.c_uint8)
VMC_REDUCE_SUM, VMC_REDUCE_MAX, VMC_REDUCE_SUM_F64 = 0, 1, 2
VMC_HOST_REDUCE_CAP_F64 = 1
# vmc_host_allreduce_fn: int hook(void* user, float* host_buf, int64_t n_elements, int32_t op)
HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, _fp, C.c_int64, C.c_int32)
# vmc_device_allreduce_fn: int hook(void* user, void* dev_buf, int64_t n_elements, int32_t op, void* stream)
DEVICE_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p)

# name -> (restype, argtypes); every symbol include/cgsvmc.h declares
SIGNATURES = {
    'vmc_num_params': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    'vmc_num_params_ansatz': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'vmc_num_params_conv': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'vmc_create': (C.c_int, [C.POINTER(VmcDesc), C.POINTER(_ctx)]),
    'vmc_destroy': (None, [_ctx]),
    'vmc_last_error': (C.c_char_p, [_ctx]),
    'vmc_set_bonds': (C.c_int, [_ctx, C.c_int32, _ip, _fp, _fp]),
    'vmc_set_params': (C.c_int, [_ctx, C.c_int, _fp]),
    'vmc_get_params': (C.c_int, [_ctx, C.c_int, _fp]),
    'vmc_transfer_params': (C.c_int, [_ctx]),
    'vmc_set_configs': (C.c_int, [_ctx, _fp]),
    'vmc_get_configs': (C.c_int, [_ctx, _fp]),
    'vmc_set_shift': (C.c_int, [_ctx, C.c_int, C.c_float]),
    'vmc_get_shift': (C.c_int, [_ctx, C.c_int, _fp]),
    'vmc_amplitude': (C.c_int, [_ctx, C.c_int, _fp, C.c_int64, _fp, _fp]),
    'vmc_mc_steps': (C.c_int, [_ctx, C.c_int64, C.POINTER(C.c_int64)]),
    'vmc_mc_step_injected': (C.c_int, [_ctx, _ip, _ip, _fp, C.POINTER(C.c_uint8)]),
    'vmc_debug_proposals': (C.c_int, [_ctx, C.c_uint64, _ip, _ip, _fp]),
    'vmc_debug_sweep_profile': (C.c_int, [_ctx, C.c_int64, C.POINTER(C.c_double)]),
    'vmc_get_step_counter': (C.c_int, [_ctx, C.POINTER(C.c_uint64)]),
    'vmc_set_step_counter': (C.c_int, [_ctx, C.c_uint64]),
    'vmc_local_energy': (C.c_int, [_ctx, C.c_int, _fp, C.POINTER(C.c_double)]),
    'vmc_local_energy_terms': (C.c_int, [_ctx, C.c_int, _fp, _fp]),
    'vmc_accumulate': (C.c_int, [_ctx, C.c_int, C.c_float]),
    'vmc_reset_accumulators': (C.c_int, [_ctx]),
    'vmc_accumulators_devptr': (C.c_int, [_ctx, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    'vmc_allreduce_accumulators': (C.c_int, [_ctx, C.c_void_p, C.c_int32]),
    'vmc_get_accumulators': (C.c_int, [_ctx, _fp]),
    'vmc_set_accumulators': (C.c_int, [_ctx, _fp]),
    'vmc_apply_adam': (C.c_int, [_ctx, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                 C.POINTER(C.c_double)]),
    'vmc_get_gradient': (C.c_int, [_ctx, C.c_int, _fp]),
    'vmc_mean_energy': (C.c_int, [_ctx, C.POINTER(C.c_double)]),
    'vmc_get_adam_state': (C.c_int, [_ctx, _fp, _fp, C.POINTER(C.c_int64)]),
    'vmc_set_adam_state': (C.c_int, [_ctx, _fp, _fp, C.c_int64]),
    'vmc_epoch_energy_gradient': (C.c_int, [_ctx, C.c_int64, C.c_int32, C.c_int64, C.c_float]),
    'vmc_epoch_log_overlap': (C.c_int, [_ctx, C.c_float, C.c_int64, C.c_int32, C.c_int64, C.c_float,
                                        C.c_float, C.c_float, C.c_float, C.c_float,
                                        C.POINTER(C.c_double)]),
    'vmc_set_host_allreduce': (C.c_int, [_ctx, HOST_ALLREDUCE_FN, C.c_void_p]),
    'vmc_set_host_allreduce_caps': (C.c_int, [_ctx, C.c_int32]),
    'vmc_set_device_allreduce': (C.c_int, [_ctx, DEVICE_ALLREDUCE_FN, C.c_void_p]),
    'vmc_evaluate': (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int64,
                               C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    'vmc_rccl_unique_id': (C.c_int, [_bp]),
    'vmc_rccl_comm_create': (C.c_int, [_bp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    'vmc_rccl_comm_destroy': (C.c_int, [C.c_void_p]),
    'vmc_rccl_last_error': (C.c_char_p, []),
    'vmc_rccl_library_path': (C.c_char_p, []),
    'vmc_device_pci_bus_id': (C.c_int, [C.c_int32, C.c_char_p, C.c_int32]),
    'vmc_hip_runtime_path': (C.c_char_p, []),
    'vmc_debug_allreduce': (C.c_int, [_ctx, C.c_void_p, C.c_int32, _fp, C.c_int64, C.c_int32]),
    'vmc_update_norm_dist': (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_float]),
    'vmc_epoch_energy_gradient_dist': (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_int64, C.c_int32,
                                                 C.c_int64, C.c_float]),
    'vmc_epoch_log_overlap_dist': (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_float, C.c_int64,
                                             C.c_int32, C.c_int64, C.c_float, C.c_float, C.c_float,
                                             C.c_float, C.c_float, C.POINTER(C.c_double)]),
    'vmc_sr_solve_dist': (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_int32,
                                    C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    'vmc_sr_reserve': (C.c_int, [_ctx, C.c_int32]),
    'vmc_sr_num_stored': (C.c_int, [_ctx, C.POINTER(C.c_int32)]),
    'vmc_sr_begin': (C.c_int, [_ctx, C.POINTER(C.c_double)]),
    'vmc_sr_matvec_partial': (C.c_int, [_ctx]),
    'vmc_sr_matvec_phase1': (C.c_int, [_ctx]),
    'vmc_sr_matvec_phase2': (C.c_int, [_ctx]),
    'vmc_sr_buffer_devptr': (C.c_int, [_ctx, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    'vmc_sr_get_buffer': (C.c_int, [_ctx, _fp]),
    'vmc_sr_set_buffer': (C.c_int, [_ctx, _fp]),
    'vmc_sr_cg_update': (C.c_int, [_ctx, C.c_float, C.POINTER(C.c_double)]),
    'vmc_sr_solve': (C.c_int, [_ctx, C.c_float, C.c_float, C.c_int32, C.POINTER(C.c_int32),
                               C.POINTER(C.c_double)]),
    'vmc_sr_get_solution': (C.c_int, [_ctx, _fp]),
    'vmc_sr_apply': (C.c_int, [_ctx, C.c_float, C.POINTER(C.c_double)]),
    'vmc_sr_debug_matvec': (C.c_int, [_ctx, _fp, C.c_float, _fp]),
    'vmc_update_norm': (C.c_int, [_ctx, C.c_float]),
    'vmc_timing_enable': (C.c_int, [_ctx, C.c_int]),
    'vmc_timing_reset': (C.c_int, [_ctx]),
    'vmc_timing_get': (C.c_int, [_ctx, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    'vmc_last_connected_rows': (C.c_int, [_ctx, C.POINTER(C.c_int64)]),
    'vmc_debug_kernel_path': (C.c_int, [_ctx, C.POINTER(C.c_int32)]),
    'vmc_debug_sweep_tile': (C.c_int, [_ctx, C.c_int32, C.POINTER(C.c_int32)]),
    'vmc_debug_conv_patch': (C.c_int, [_ctx, C.c_int64, C.POINTER(C.c_int32)]),
    'vmc_synchronize': (C.c_int, [_ctx]),
    'vmc_debug_gemm': (C.c_int, [_ctx, C.c_int32, C.c_int32, C.c_int32, _fp, C.c_int64, C.c_int64,
                                 C.c_int64, _fp, C.c_int64, C.c_int64, C.c_int64, _fp]),
}

_lib = None
_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
_STAMP_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libcgsvmc_hip.stamp')
# the files the library is built from, in the order csrc/Makefile hashes them
_SOURCES = ('vmc_api.hip', 'vmc_api_cgen.hip', 'vmc_api_sweep.hip', 'vmc_api_train.hip', 'vmc_api_coll.hip', 'vmc_api_sr.hip', 'mlp.hip', 'eloc.hip', 'grad.hip', 'sr.hip', 'srmm.hip', 'conv.hip', 'conv32.hip', 'conv48.hip', 'conv64.hip', 'conv_general.hip', 'conv_band.hip', 'conv_patch.hip', 'wide.hip', 'tail_split.hip', 'sweep_split.hip', 'sweep8.hip', 'act_tail.hip',
            'act_sweep.hip', 'vmc_ctx.hpp', 'plan.hpp', 'common.hpp', 'tail16.hpp', 'tail_lds.hpp', 'sweep16.hpp', 'conv.hpp', 'conv_kernels.hpp',
            'conv_wide.hpp',
            os.path.join('..', '..', 'include', 'cgsvmc.h'), 'Makefile')


def library_path() -> str:
  return _LIB_PATH


def source_hash() -> str:
  """sha256 over the library's sources (same byte stream as `cat ... | sha256sum` in the
  Makefile)."""
  import hashlib
  h = hashlib.sha256()
  for name in _SOURCES:
    with open(os.path.join(_CSRC, name), 'rb') as f:
      h.update(f.read())
  return h.hexdigest()


def stamp_matches() -> bool:
  """True when libcgsvmc_hip.so was linked from exactly the sources in the tree, with the
  product flags (a build with EXTRA=... flags -- diagnostics such as the s_memtime-stamped sampler
  -- only loads under CGS_VMC_ALLOW_EXTRA_BUILD=1)."""
  try:
    with open(_STAMP_PATH) as f:
      fields = f.read().split()
    if fields[0] != source_hash():
      return False
    extra = fields[1] if len(fields) > 1 else '-'
    return extra == '-' or os.environ.get('CGS_VMC_ALLOW_EXTRA_BUILD', '0') == '1'
  except (OSError, IndexError):
    return False


def load():
  """Loads the shared library and binds every declared symbol (no GPU needed)."""
  global _lib
  if _lib is not None:
    return _lib
  # Diagnostic builds (in-kernel stamps, ...) live under a path of their own and are selected explicitly:
  # CGS_VMC_DIAGNOSTIC_LIBRARY=/path/lib.so together with CGS_VMC_ALLOW_EXTRA_BUILD=1.  They never replace
  # the product library in the tree (tools/wgrad_stamps.sh).
  diag = os.environ.get('CGS_VMC_DIAGNOSTIC_LIBRARY')
  if diag:
    if os.environ.get('CGS_VMC_ALLOW_EXTRA_BUILD', '0') != '1':
      raise HipLibraryError('CGS_VMC_DIAGNOSTIC_LIBRARY is set without CGS_VMC_ALLOW_EXTRA_BUILD=1')
    import sys
    sys.stderr.write('cgs_vmc_amd: loading the DIAGNOSTIC library {} (never quote its run times)\n'.format(diag))
    return _bind(diag, preload_torch=True)
  if not os.path.exists(_LIB_PATH):
    raise HipLibraryError(
        '{} not found: build it with __graft_entry__.build() (hipcc --offload-arch=gfx950). '
        'The VMC hot path has no CPU fallback.'.format(_LIB_PATH))
  if not stamp_matches():
    raise HipLibraryError(
        '{} was not built from the sources in {} (stamp mismatch): rebuild it with '
        '__graft_entry__.build() or `make -C cgs_vmc_amd/csrc`.'.format(_LIB_PATH, _CSRC))
  return _bind(_LIB_PATH, preload_torch=True)


def _bind(path, preload_torch):
  global _lib
  # ONE HIP runtime per process.  torch bundles its own libamdhip64 (same soname as ROCm's): whichever is
  # loaded first serves this library, and torch always loads its own -- so with this library first the
  # process ends up with two runtimes (engines on one, torch on the other: "No HIP GPUs are available"
  # from torch's late initialisation and aborts at exit have both been seen that way).  Loading torch
  # first, where it is installed, makes every import order end in torch's single runtime.
  # CGS_VMC_NO_TORCH_PRELOAD=1 skips it (torch-free hosts that want the seconds of start-up back).
  if preload_torch and os.environ.get('CGS_VMC_NO_TORCH_PRELOAD', '0') != '1':
    try:
      import torch  # noqa: F401  pylint: disable=unused-import,import-outside-toplevel
    except ImportError:
      pass
    except Exception as e:  # pylint: disable=broad-except
      # a broken torch install (OSError / RuntimeError from its own shared objects) must not make
      # the library unloadable for torch-free use
      import warnings
      warnings.warn('cgs_vmc_amd: importing torch before libcgsvmc_hip.so failed ({!r}); loading the '
                    'library on the system HIP runtime'.format(e))
  try:
    lib = C.CDLL(path)
  except OSError as e:
    raise HipLibraryError('cannot load {}: {}'.format(path, e)) from e
  for name, (res, args) in SIGNATURES.items():
    try:
      fn = getattr(lib, name)
    except AttributeError as e:
      raise HipLibraryError('{} does not export {}'.format(path, name)) from e
    fn.restype = res
    fn.argtypes = args
  _lib = lib
  return lib
