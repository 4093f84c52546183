"""Wavefunction interface and the fully-connected ansatz (mirror of
cgs_vmc/wavefunctions.py, hot-path scope: SURVEY.md 8a rows a6-a10).

A `Wavefunction` is a host-side description (shape, activations, parameter vector).  The
arithmetic runs in libcgsvmc_hip.so: applying a wavefunction to the CONFIGS variable of
graph_builders binds it to that variable's `VmcEngine` as parameter set psi (the first
wavefunction bound) or omega (its deep copy, the LogOverlapITSWO supervisor).
"""
from __future__ import annotations

import copy
import inspect
import os
from typing import Any, Dict, List, Optional

import numpy as np

from . import _hip
from . import layers
from . import session as session_lib

_name_counts: Dict[str, int] = {}
_init_counter = [0]


def _init_count() -> int:
  _init_counter[0] += 1
  return _init_counter[0] - 1


def _unique_name(name: str) -> str:
  """Sonnet module name uniquification: name, name_1, name_2, ..."""
  n = _name_counts.get(name, 0)
  _name_counts[name] = n + 1
  return name if n == 0 else '%s_%d' % (name, n)


def reset_name_scope():
  _name_counts.clear()
  _init_counter[0] = 0


class Wavefunction:
  """Wavefunction interface (wavefunctions.py:21-297)."""

  def __init__(self, name: str = 'wavefunction'):
    self._name = name
    self._unique_name = _unique_name(name)
    self._sub_wavefunctions: List['Wavefunction'] = []
    self._exp_norm_shift = None
    self._engine = None
    self._which = None

  # -- graph connection ----------------------------------------------------
  def __call__(self, inputs) -> session_lib.Tensor:
    return self._build(inputs)

  def _build(self, inputs):
    raise NotImplementedError

  def __add__(self, other):
    raise NotImplementedError('sum/diff/prod composites are outside the MI355X hot path '
                              '(SURVEY.md 2: composites OUT OF SCOPE)')

  __mul__ = __add__
  __sub__ = __add__

  def get_trainable_variables(self) -> List[session_lib.Variable]:
    """wavefunctions.py:167-175: own variables in creation order, then sub-wavefunctions'."""
    variables = list(self._own_variables())
    for sub in self._sub_wavefunctions:
      variables += sub.get_trainable_variables()
    return variables

  def _own_variables(self):
    return []

  def __deepcopy__(self, memo: Dict[int, Any]) -> 'Wavefunction':
    """wavefunctions.py:177-204: a twin built from the same constructor arguments (deep-copied,
    so sub-wavefunctions get twins too) with fresh variables under the name dc_<name>."""
    twin = memo.get(id(self))
    if twin is None:
      ctor = inspect.signature(type(self).__init__).parameters
      kwargs = {arg: copy.deepcopy(getattr(self, '_' + arg), memo)
                for arg in ctor if arg not in ('self', 'name')}
      twin = type(self)(name='dc_' + self._unique_name, **kwargs)
      memo[id(self)] = twin
    return twin

  # -- normalisation -------------------------------------------------------
  def add_exp_normalization(self, initial_exp_norm_shift: float = -10.):
    """wavefunctions.py:206-232: non-trainable scalar shift, psi = exp(logit - shift)."""
    self._exp_norm_shift = np.float32(initial_exp_norm_shift)

  def normalize_batch(self, batch_of_amplitudes, max_value: float = 1e10):
    """wavefunctions.py:234-259."""
    if self._exp_norm_shift is None:
      return None

    def run():
      log_max = self._global_log_max(batch_of_amplitudes)
      self._set_shift(np.float32(self._get_shift() + (log_max - np.log(np.float32(max_value)))))
    return session_lib.Op(run, 'normalize_batch')

  def _global_log_max(self, batch_of_amplitudes) -> np.float32:
    """log(max_b psi_b) over ALL ranks (wavefunctions.py:250, 283).  For this ansatz's own
    amplitudes the max is taken over the logits and `log(exp(max_logit - shift))` is evaluated
    once, as vmc_update_norm does: where psi overflows float32 the reference's value is inf;
    the logit-domain value max_logit - shift is used there, on every path (single GPU, sharded,
    op-by-op), so that sharded and unsharded runs agree."""
    from . import parallel
    own = (isinstance(batch_of_amplitudes, AmplitudeTensor)
           and batch_of_amplitudes.wavefunction is self and self._exp_norm_shift is not None)
    if own:
      top = np.float32(parallel.allreduce_max(float(np.max(batch_of_amplitudes.logits()))))
      gap = np.float32(top - np.float32(self._get_shift()))
      with np.errstate(over='ignore'):
        psi_max = np.exp(gap, dtype=np.float32)
      if np.isfinite(psi_max) and psi_max > 0:
        return np.float32(np.log(psi_max))
      return gap
    psi = np.asarray(batch_of_amplitudes._run())
    with np.errstate(divide='ignore'):
      return np.float32(np.log(np.float32(parallel.allreduce_max(float(np.max(psi))))))

  def update_norm(self, batch_of_amplitudes, max_value: float = 1e10):
    """wavefunctions.py:261-288."""
    if self._exp_norm_shift is None:
      return None
    return session_lib.Op(lambda: self._update_norm(batch_of_amplitudes, max_value),
                          'update_norm')

  def _update_norm(self, batch_of_amplitudes, max_value):
    raise NotImplementedError

  def _get_shift(self):
    return self._exp_norm_shift

  def _set_shift(self, value):
    self._exp_norm_shift = np.float32(value)

  @classmethod
  def from_hparams(cls, hparams, name: str = '') -> 'Wavefunction':
    raise NotImplementedError


def module_transfer_ops(source_module: Wavefunction, target_module: Wavefunction) -> session_lib.Op:
  """wavefunctions.py:300-325: assign every trainable variable of source to target."""
  def run():
    if (source_module._engine is not None and source_module._engine is target_module._engine
        and source_module._which == _hip.VMC_PSI and target_module._which == _hip.VMC_OMEGA):
      source_module._engine.transfer_params()          # device-to-device
      target_module._has_values, target_module._theta = True, None
      return
    src = source_module.get_trainable_variables()
    dst = target_module.get_trainable_variables()
    for s, t in zip(src, dst):
      t.load(s.eval())
  return session_lib.Op(run, 'module_transfer')


class FullyConnectedNetwork(Wavefunction):
  """[Linear(layer_size), nonlinearity] x num_layers -> Linear(1) -> squeeze ->
  (- exp_norm_shift) -> exp   (wavefunctions.py:328-388)."""
  _ansatz = 'fully_connected'     # engine kernel family (include/cgsvmc.h VMC_ANSATZ_*)

  def __init__(self, num_layers: int, layer_size: int,
               nonlinearity=layers.NONLINEARITIES['relu'],
               output_activation=layers.NONLINEARITIES['exp'],
               name: str = 'fully_connected_network'):
    super(FullyConnectedNetwork, self).__init__(name=name)
    self._num_layers = num_layers
    self._layer_size = layer_size
    self._nonlinearity = nonlinearity
    self._output_activation = output_activation
    if output_activation == layers.NONLINEARITIES['exp']:
      self.add_exp_normalization()
    self._n_sites: Optional[int] = None
    self._theta: Optional[np.ndarray] = None     # host copy until bound to an engine
    self._has_values = False                     # variables initialised / restored
    session_lib.get_default_graph().global_initializers.append(self._maybe_initialize)

  # -- parameters ----------------------------------------------------------
  def _shapes(self):
    shapes, names = [], []
    fan_in = self._n_sites
    for l in range(self._num_layers + 1):
      out = self._layer_size if l < self._num_layers else 1
      lin = 'linear' if l == 0 else 'linear_%d' % l
      names += ['%s/%s/w' % (self._unique_name, lin), '%s/%s/b' % (self._unique_name, lin)]
      shapes += [(fan_in, out), (out,)]
      fan_in = out
    return names, shapes

  @property
  def num_params(self) -> int:
    return int(sum(int(np.prod(s)) for s in self._shapes()[1]))

  def _maybe_initialize(self):
    if self._n_sites is not None and self._get_theta(allow_none=True) is None:
      # unseeded like the reference unless CGS_VMC_INIT_SEED is set (tests, reproducible runs)
      seed = os.environ.get('CGS_VMC_INIT_SEED')
      self.initialize(None if seed is None else int(seed) + _init_count())

  def initialize(self, seed=None):
    """snt.Linear defaults: w ~ truncated normal(sigma = 1/sqrt(fan_in)), b = 0."""
    if self._n_sites is None:
      raise ValueError('wavefunction is not connected to inputs yet')
    rng = np.random.default_rng(seed)
    parts = []
    for shp in self._shapes()[1]:
      if len(shp) == 2:
        w = rng.standard_normal(shp)
        bad = np.abs(w) > 2
        while bad.any():
          w[bad] = rng.standard_normal(int(bad.sum()))
          bad = np.abs(w) > 2
        parts.append((w / np.sqrt(shp[0])).ravel())
      else:
        parts.append(np.zeros(shp).ravel())
    self._set_theta(np.concatenate(parts).astype(np.float32))

  def _get_theta(self, allow_none=False):
    if not self._has_values:
      if allow_none:
        return None
      raise ValueError('Attempting to use uninitialized variables of %s' % self._unique_name)
    if self._engine is not None and self._theta is None:
      return self._engine.get_params(self._which)
    return self._theta

  def _set_theta(self, theta):
    theta = np.ascontiguousarray(theta, np.float32)
    self._has_values = True
    if self._engine is not None:
      self._engine.set_params(theta, self._which)
      self._theta = None          # the device copy is authoritative from now on
    else:
      self._theta = theta

  def _own_variables(self):
    if self._n_sites is None:
      raise ValueError('wavefunction %s has no variables before it is connected to inputs'
                       % self._unique_name)
    names, shapes = self._shapes()
    out, off = [], 0
    for name, shp in zip(names, shapes):
      n = int(np.prod(shp))

      def getter(off=off, n=n):
        return self._get_theta()[off:off + n]

      def setter(value, off=off, n=n):
        theta = self._get_theta(allow_none=True)
        if theta is None:
          theta = np.zeros(self.num_params, np.float32)
        theta = theta.copy()
        theta[off:off + n] = np.asarray(value, np.float32).ravel()
        self._set_theta(theta)

      out.append(session_lib.Variable(name, shp, getter, setter, trainable=True))
      off += n
    return out

  # -- engine binding ------------------------------------------------------
  def _engine_spec(self):
    """Keyword arguments that describe this ansatz to VmcEngine / vmc_create."""
    return dict(ansatz=self._ansatz, num_layers=self._num_layers, layer_size=self._layer_size,
                nonlinearity=self._nonlinearity.name,
                output_activation=self._output_activation.name)

  def _bind(self, configs_var):
    """Connects this ansatz to the engine that owns `configs_var`."""
    n_sites = configs_var.shape[1]
    if self._n_sites is not None and self._n_sites != n_sites:
      raise ValueError('Input tensor has wrong shape.')
    self._n_sites = n_sites
    engine = configs_var._get_engine(self)
    if self._engine is engine:
      return engine
    if self._engine is not None:
      raise ValueError('wavefunction %s is already bound to another CONFIGS variable'
                       % self._unique_name)
    which = configs_var._claim_slot(self)
    host_theta = self._theta
    self._engine, self._which = engine, which
    if host_theta is not None:
      engine.set_params(host_theta, which)
      self._theta = None
    if self._exp_norm_shift is not None:
      engine.set_shift(float(self._exp_norm_shift), which)
    return engine

  def _get_shift(self):
    if self._engine is not None:
      return np.float32(self._engine.get_shift(self._which))
    return self._exp_norm_shift

  def _set_shift(self, value):
    self._exp_norm_shift = np.float32(value)
    if self._engine is not None:
      self._engine.set_shift(float(value), self._which)

  def _build(self, inputs) -> session_lib.Tensor:
    """wavefunctions.py:355-371."""
    from . import graph_builders
    if isinstance(inputs, graph_builders.ConfigsVariable):
      engine = self._bind(inputs)
      return AmplitudeTensor(self, engine, None)
    arr = np.asarray(inputs, np.float32)
    if arr.ndim != 2 or (self._n_sites is not None and arr.shape[1] != self._n_sites):
      raise ValueError('Input tensor has wrong shape.')
    if self._engine is None:
      raise ValueError('apply the wavefunction to the CONFIGS variable first '
                       '(graph_builders.get_configs) so that it is bound to a GPU engine')
    return AmplitudeTensor(self, self._engine, arr)

  def _update_norm(self, batch_of_amplitudes, max_value):
    from . import parallel
    if (isinstance(batch_of_amplitudes, AmplitudeTensor) and batch_of_amplitudes.configs is None
        and batch_of_amplitudes.wavefunction is self):
      if parallel.world_size() == 1:
        self._engine.update_norm(max_value)        # max-reduce on the GPU
      else:                                        # ... and a one-float MAX all-reduce in stream
        self._engine.update_norm_dist(parallel.collective(), max_value)
      return
    log_max = self._global_log_max(batch_of_amplitudes)
    max_log = np.log(np.float32(max_value))
    if log_max > max_log:
      self._set_shift(np.float32(self._get_shift() + (log_max - max_log)))

  @classmethod
  def from_hparams(cls, hparams, name: str = '') -> 'Wavefunction':
    """wavefunctions.py:373-388."""
    fcnn_params = {
        'num_layers': hparams.num_fc_layers,
        'layer_size': hparams.fc_layer_size,
        'output_activation': layers.NONLINEARITIES[hparams.output_activation],
        'nonlinearity': layers.NONLINEARITIES[hparams.nonlinearity],
    }
    if name:
      fcnn_params['name'] = name
    return cls(**fcnn_params)


class RestrictedBoltzmannNetwork(FullyConnectedNetwork):
  """Extended restricted Boltzmann machine (wavefunctions.py:391-452):
  psi = exp(onsite(x) + sum_h log cosh(Linear(H)([Linear(H), nonlinearity] x num_layers (x)))_h
            - exp_norm_shift),  onsite = Linear(1).
  num_layers = 0 is the classic RBM.  Same engine, same kernels as the fully-connected ansatz
  with a log-cosh output epilogue and the rank-2 onsite update (csrc/mlp.hip, RBM variants)."""
  _ansatz = 'rbm'

  def __init__(self, num_layers: int, layer_size: int,
               nonlinearity=layers.NONLINEARITIES['relu'],
               name: str = 'restricted_boltzmann_network'):
    super(RestrictedBoltzmannNetwork, self).__init__(
        num_layers=num_layers, layer_size=layer_size, nonlinearity=nonlinearity,
        output_activation=layers.NONLINEARITIES['exp'], name=name)

  def _shapes(self):
    """Creation order of the snt.Linear variables: Sonnet v1 creates them when a module is first
    connected, and _build (wavefunctions.py:436-437) connects the onsite layer -- the LAST
    module constructed, hence `linear_{num_layers+1}` -- before the Sequential."""
    n, h, L = self._n_sites, self._layer_size, self._num_layers
    u = self._unique_name
    names = ['%s/linear_%d/w' % (u, L + 1), '%s/linear_%d/b' % (u, L + 1)]
    shapes = [(n, 1), (1,)]
    fan_in = n
    for l in range(L + 1):
      lin = 'linear' if l == 0 else 'linear_%d' % l
      names += ['%s/%s/w' % (u, lin), '%s/%s/b' % (u, lin)]
      shapes += [(fan_in, h), (h,)]
      fan_in = h
    return names, shapes

  @classmethod
  def from_hparams(cls, hparams, name: str = '') -> 'Wavefunction':
    """wavefunctions.py:440-452."""
    rbm_params = {
        'num_layers': hparams.num_fc_layers,
        'layer_size': hparams.fc_layer_size,
        'nonlinearity': layers.NONLINEARITIES[hparams.nonlinearity],
    }
    if name:
      rbm_params['name'] = name
    return cls(**rbm_params)


class Conv2DNetwork(FullyConnectedNetwork):
  """[Conv2dPeriodic(num_filters, kernel_size), nonlinearity] x (num_layers - 1),
  Conv2dPeriodic, reduce_sum over sites and channels, (- exp_norm_shift), exp
  (wavefunctions.py:531-615; layers.Conv2dPeriodic, layers.py:89-160).  Inputs are reshaped to
  [-1, size_x, size_y, 1]; the kernels are the periodic implicit-GEMM family of csrc/conv.hip."""
  _ansatz = 'conv_2d'

  def __init__(self, num_layers: int, num_filters: int, kernel_size: int, size_x: int,
               size_y: int, nonlinearity=layers.NONLINEARITIES['relu'],
               output_activation=layers.NONLINEARITIES['exp'], name: str = 'conv_2d_network'):
    super(Conv2DNetwork, self).__init__(
        num_layers=num_layers, layer_size=num_filters, nonlinearity=nonlinearity,
        output_activation=output_activation, name=name)
    self._num_filters = num_filters
    self._kernel_size = kernel_size
    self._size_x = size_x
    self._size_y = size_y

  def _conv_scopes(self):
    """Variable scopes of the snt.Conv2D modules in connection order."""
    return ['conv_2d_periodic' if l == 0 else 'conv_2d_periodic_%d' % l
            for l in range(self._num_layers)]

  def _shapes(self):
    if self._n_sites != self._size_x * self._size_y:
      raise ValueError('Input tensor has wrong shape.')        # tf.reshape fails in the reference
    k, f, u = self._kernel_size, self._num_filters, self._unique_name
    names, shapes, cin = [], [], 1
    for scope in self._conv_scopes():
      names += ['%s/%s/conv_2d/w' % (u, scope), '%s/%s/conv_2d/b' % (u, scope)]
      shapes += [(k, k, cin, f), (f,)]
      cin = f
    return names, shapes

  def initialize(self, seed=None):
    """snt.Conv2D defaults: w ~ truncated normal(sigma = 1/sqrt(k*k*in_channels)), b = 0."""
    if self._n_sites is None:
      raise ValueError('wavefunction is not connected to inputs yet')
    rng = np.random.default_rng(seed)
    parts = []
    for shp in self._shapes()[1]:
      if len(shp) == 4:
        w = rng.standard_normal(shp)
        bad = np.abs(w) > 2
        while bad.any():
          w[bad] = rng.standard_normal(int(bad.sum()))
          bad = np.abs(w) > 2
        parts.append((w / np.sqrt(shp[0] * shp[1] * shp[2])).ravel())
      else:
        parts.append(np.zeros(shp).ravel())
    self._set_theta(np.concatenate(parts).astype(np.float32))

  def _engine_spec(self):
    spec = super(Conv2DNetwork, self)._engine_spec()
    spec.update(kernel_size=self._kernel_size, size_x=self._size_x, size_y=self._size_y)
    return spec

  @classmethod
  def from_hparams(cls, hparams, name: str = '') -> 'Wavefunction':
    """wavefunctions.py:600-615."""
    conv_2d_params = {
        'num_layers': hparams.num_conv_layers,
        'num_filters': hparams.num_conv_filters,
        'kernel_size': hparams.kernel_size,
        'size_x': hparams.size_x,
        'size_y': hparams.size_y,
        'output_activation': layers.NONLINEARITIES[hparams.output_activation],
        'nonlinearity': layers.NONLINEARITIES[hparams.nonlinearity],
    }
    if name:
      conv_2d_params['name'] = name
    return cls(**conv_2d_params)


class ResNet2D(Conv2DNetwork):
  """Conv2dPeriodic, then num_blocks x ResBlock2d (x + conv(selu(conv(x)))), reduce_sum,
  (- exp_norm_shift), exp   (wavefunctions.py:710-809; layers.ResBlock2d, layers.py:163-229).
  The reference only has the plain block for 2D (bottleneck=True names a class layers.py does not
  define) and a block only type-checks at stride 1, so those are the supported settings."""
  _ansatz = 'res_net_2d'

  def __init__(self, num_blocks: int, num_filters: int, kernel_size: int, conv_stride: int,
               size_x: int, size_y: int, bottleneck: bool = False,
               output_activation=layers.NONLINEARITIES['exp'], name: str = 'res_net_2d'):
    if bottleneck:
      raise AttributeError("module 'layers' has no attribute 'BottleneckResBlock2d'")
    if conv_stride != 1:
      raise ValueError('Inputs shape is not compatable with filters.')   # layers.py:218-219
    super(ResNet2D, self).__init__(
        num_layers=num_blocks, num_filters=num_filters, kernel_size=kernel_size, size_x=size_x,
        size_y=size_y, nonlinearity=layers.NONLINEARITIES['relu'],
        output_activation=output_activation, name=name)
    self._num_blocks = num_blocks
    self._conv_stride = conv_stride
    self._bottleneck = bottleneck

  def _conv_scopes(self):
    scopes = ['conv_2d_periodic']
    for blk in range(self._num_blocks):
      block = 'res_block_2d' if blk == 0 else 'res_block_2d_%d' % blk
      scopes += ['%s/first_conv' % block, '%s/second_conv' % block]
    return scopes

  @classmethod
  def from_hparams(cls, hparams, name: str = '') -> 'Wavefunction':
    """wavefunctions.py:794-809."""
    res_net_2d_params = {
        'num_blocks': hparams.num_resnet_blocks,
        'num_filters': hparams.num_conv_filters,
        'kernel_size': hparams.kernel_size,
        'conv_stride': hparams.conv_strides,
        'size_x': hparams.size_x,
        'size_y': hparams.size_y,
        'output_activation': layers.NONLINEARITIES[hparams.output_activation],
    }
    if name:
      res_net_2d_params['name'] = name
    return cls(**res_net_2d_params)


class Conv1DNetwork(Conv2DNetwork):
  """[Conv1dPeriodic(num_filters, kernel_size), nonlinearity] x (num_layers - 1), Conv1dPeriodic,
  reduce_sum over sites and channels, (- exp_norm_shift), exp   (wavefunctions.py:455-527;
  layers.Conv1dPeriodic, layers.py:24-86).  Inputs are expanded to [B, N, 1]; the kernels are those
  of the 2-D types on an N x 1 lattice with k x 1 taps and the 1-D padding rule (an even kernel
  pads k/2 in front and k/2 - 1 behind, layers.py:66-72 -- the mirror image of the 2-D module)."""
  _ansatz = 'conv_1d'

  def __init__(self, num_layers: int, num_filters: int, kernel_size: int,
               nonlinearity=layers.NONLINEARITIES['relu'],
               output_activation=layers.NONLINEARITIES['exp'], name: str = 'conv_1d_network'):
    super(Conv1DNetwork, self).__init__(
        num_layers=num_layers, num_filters=num_filters, kernel_size=kernel_size, size_x=0, size_y=1,
        nonlinearity=nonlinearity, output_activation=output_activation, name=name)

  def _conv_scopes(self):
    return ['conv_1d_periodic' if l == 0 else 'conv_1d_periodic_%d' % l
            for l in range(self._num_layers)]

  def _shapes(self):
    k, f, u = self._kernel_size, self._num_filters, self._unique_name
    names, shapes, cin = [], [], 1
    for scope in self._conv_scopes():
      names += ['%s/%s/conv_1d/w' % (u, scope), '%s/%s/conv_1d/b' % (u, scope)]
      shapes += [(k, cin, f), (f,)]
      cin = f
    return names, shapes

  def initialize(self, seed=None):
    """snt.Conv1D defaults: w ~ truncated normal(sigma = 1/sqrt(k*in_channels)), b = 0."""
    if self._n_sites is None:
      raise ValueError('wavefunction is not connected to inputs yet')
    rng = np.random.default_rng(seed)
    parts = []
    for shp in self._shapes()[1]:
      if len(shp) == 3:
        w = rng.standard_normal(shp)
        bad = np.abs(w) > 2
        while bad.any():
          w[bad] = rng.standard_normal(int(bad.sum()))
          bad = np.abs(w) > 2
        parts.append((w / np.sqrt(shp[0] * shp[1])).ravel())
      else:
        parts.append(np.zeros(shp).ravel())
    self._set_theta(np.concatenate(parts).astype(np.float32))

  def _engine_spec(self):
    spec = FullyConnectedNetwork._engine_spec(self)
    spec.update(kernel_size=self._kernel_size, size_x=0, size_y=0)
    return spec

  @classmethod
  def from_hparams(cls, hparams, name: str = '') -> 'Wavefunction':
    """wavefunctions.py:513-527."""
    conv_1d_params = {
        'num_layers': hparams.num_conv_layers,
        'num_filters': hparams.num_conv_filters,
        'kernel_size': hparams.kernel_size,
        'output_activation': layers.NONLINEARITIES[hparams.output_activation],
        'nonlinearity': layers.NONLINEARITIES[hparams.nonlinearity],
    }
    if name:
      conv_1d_params['name'] = name
    return cls(**conv_1d_params)


class ResNet1D(Conv1DNetwork):
  """Conv1dPeriodic, then num_blocks x ResBlock1d (x + conv(selu(conv(x))), layers.py:290-293),
  reduce_sum, (- exp_norm_shift), exp   (wavefunctions.py:618-707).  Plain blocks at stride 1 are
  supported (a strided block does not type-check against its shortcut, layers.py:281-282;
  bottleneck blocks are outside the MI355X hot path)."""
  _ansatz = 'res_net_1d'

  def __init__(self, num_blocks: int, num_filters: int, kernel_size: int, conv_stride: int,
               bottleneck: bool = False, output_activation=layers.NONLINEARITIES['exp'],
               name: str = 'res_net_1d'):
    if bottleneck:
      raise NotImplementedError('BottleneckResBlock1d is outside the MI355X hot path')
    if conv_stride != 1:
      raise ValueError('Inputs shape is not compatable with filters.')
    super(ResNet1D, self).__init__(
        num_layers=num_blocks, num_filters=num_filters, kernel_size=kernel_size,
        nonlinearity=layers.NONLINEARITIES['relu'], output_activation=output_activation, name=name)
    self._num_blocks = num_blocks
    self._conv_stride = conv_stride
    self._bottleneck = bottleneck

  def _conv_scopes(self):
    scopes = ['conv_1d_periodic']
    for blk in range(self._num_blocks):
      block = 'res_block_1d' if blk == 0 else 'res_block_1d_%d' % blk
      scopes += ['%s/first_conv' % block, '%s/second_conv' % block]
    return scopes

  @classmethod
  def from_hparams(cls, hparams, name: str = '') -> 'Wavefunction':
    """wavefunctions.py:692-707."""
    res_net_1d_params = {
        'num_blocks': hparams.num_resnet_blocks,
        'num_filters': hparams.num_conv_filters,
        'kernel_size': hparams.kernel_size,
        'conv_stride': hparams.conv_strides,
        'output_activation': layers.NONLINEARITIES[hparams.output_activation],
    }
    if name:
      res_net_1d_params['name'] = name
    return cls(**res_net_1d_params)


class AmplitudeTensor(session_lib.Tensor):
  """psi = wavefunction(inputs); evaluates to a float32 array [rows]."""

  def __init__(self, wavefunction, engine, configs):
    self.wavefunction = wavefunction
    self.engine = engine
    self.configs = configs
    super(AmplitudeTensor, self).__init__(self._value, 'psi')

  def _value(self):
    return self.engine.amplitude(self.configs, self.wavefunction._which)[1]

  def logits(self):
    return self.engine.amplitude(self.configs, self.wavefunction._which)[0]


class _OutOfScope(Wavefunction):
  """Registered ansatz names whose kernels are not part of the MI355X hot path."""
  _kind = ''

  @classmethod
  def from_hparams(cls, hparams, name: str = ''):
    raise NotImplementedError(
        "wavefunction_type '%s' is outside the MI355X hot path (SURVEY.md 2); only "
        "'fully_connected', 'rbm', 'conv_1d', 'conv_2d', 'res_net_1d' and 'res_net_2d' have HIP "
        "kernels" % cls._kind)


def _stub(kind):
  return type('OutOfScope_' + kind, (_OutOfScope,), {'_kind': kind})


def build_wavefunction(hparams) -> Wavefunction:
  """wavefunctions.py:1157-1196."""
  wavefunction_type = hparams.wavefunction_type
  if wavefunction_type in WAVEFUNCTION_TYPES:
    return WAVEFUNCTION_TYPES[wavefunction_type].from_hparams(hparams)
  if hparams.wavefunction_type in ('sum', 'diff', 'prod'):
    raise NotImplementedError('composite wavefunctions are outside the MI355X hot path')
  raise ValueError('Provided wavefunction_type is not registered.')


WAVEFUNCTION_TYPES = {
    'fully_connected': FullyConnectedNetwork,
    'rbm': RestrictedBoltzmannNetwork,
    'conv_1d': Conv1DNetwork,
    'conv_2d': Conv2DNetwork,
    'mps': _stub('mps'),
    'pbdg': _stub('pbdg'),
    'fully_connected_nnb': _stub('fully_connected_nnb'),
    'res_net_1d': ResNet1D,
    'res_net_2d': ResNet2D,
    'ed_vector': _stub('ed_vector'),
    'gnn': _stub('gnn'),
}
