"""Hyper-parameters and initial configurations (mirror of cgs_vmc/utils.py).

`HParams` restates the parts of tf.contrib.training.HParams the reference uses
(attribute access, set_hparam, parse, override_from_dict, to_proto text) and
`load_hparams` reads the text-proto `hparams.pbtxt` the reference writes
(run_training.py:100-101, utils.py:153-166) without TensorFlow or protobuf.
"""
from __future__ import annotations

import copy
import re
from typing import Any, Dict

import numpy as np


class HParams:
  """Minimal tf.contrib.training.HParams: typed name -> value store."""

  def __init__(self, **kwargs: Any):
    object.__setattr__(self, '_hparam_types', {})
    for k, v in kwargs.items():
      self.add_hparam(k, v)

  # -- construction ---------------------------------------------------------
  def add_hparam(self, name: str, value: Any):
    if name in self._hparam_types:
      raise ValueError('Hyperparameter name is reserved: %s' % name)
    if isinstance(value, (list, tuple)):
      if not value:
        raise ValueError('Multi-valued hyperparameters cannot be empty: %s' % name)
      self._hparam_types[name] = (type(value[0]), True)
      value = list(value)
    else:
      self._hparam_types[name] = (type(value), False)
    object.__setattr__(self, name, value)

  def __setattr__(self, name, value):
    if name.startswith('_') or name not in self._hparam_types:
      object.__setattr__(self, name, value)
    else:
      self.set_hparam(name, value)

  @staticmethod
  def _cast(value, typ, name):
    if typ is bool:
      if isinstance(value, str):
        if value.lower() in ('true', '1'):
          return True
        if value.lower() in ('false', '0'):
          return False
        raise ValueError('Could not parse bool for %s: %r' % (name, value))
      return bool(value)
    if typ is int:
      if isinstance(value, float) and value != int(value):
        raise ValueError('Must pass an int for %s, got %r' % (name, value))
      return int(value)
    if typ is float:
      return float(value)
    if typ is str:
      return str(value)
    return typ(value)

  def set_hparam(self, name: str, value: Any):
    if name not in self._hparam_types:
      raise KeyError('Unknown hyperparameter: %s' % name)
    typ, is_list = self._hparam_types[name]
    if isinstance(value, (list, tuple)):
      if not is_list:
        raise ValueError('Must not pass a list for single-valued parameter: %s' % name)
      object.__setattr__(self, name, [self._cast(v, typ, name) for v in value])
    else:
      if is_list:
        raise ValueError('Must pass a list for multi-valued parameter: %s.' % name)
      object.__setattr__(self, name, self._cast(value, typ, name))

  def override_from_dict(self, values_dict: Dict[str, Any]):
    for name, value in values_dict.items():
      self.set_hparam(name, value)
    return self

  def parse(self, values: str):
    """Overrides from 'a=1,b=[1,2],c=name' (tf HParams.parse syntax)."""
    if not values:
      return self
    pos = 0
    pattern = re.compile(r'\s*([A-Za-z_][A-Za-z0-9_]*)\s*=\s*(\[[^\]]*\]|[^,\[\]]*)\s*(,|$)')
    while pos < len(values):
      m = pattern.match(values, pos)
      if not m:
        raise ValueError('Malformed hyperparameter value: %s' % values[pos:])
      name, val = m.group(1), m.group(2).strip()
      if name not in self._hparam_types:
        raise ValueError('Unknown hyperparameter type for %s' % name)
      if val.startswith('['):
        items = [v.strip().strip('\'"') for v in val[1:-1].split(',') if v.strip()]
        self.set_hparam(name, items)
      else:
        self.set_hparam(name, val.strip('\'"'))
      pos = m.end()
    return self

  def values(self) -> Dict[str, Any]:
    return {n: getattr(self, n) for n in self._hparam_types}

  def __contains__(self, key):
    return key in self._hparam_types

  def __copy__(self):
    new = HParams()
    for n in self._hparam_types:
      new.add_hparam(n, copy.copy(getattr(self, n)))
    return new

  def __repr__(self):
    return 'HParams(%s)' % ', '.join('%s=%r' % kv for kv in sorted(self.values().items()))

  # -- text proto (HParamDef) ----------------------------------------------
  def to_proto(self) -> 'HParamDefText':
    return HParamDefText(self)


def _proto_scalar(typ, v):
  if typ is bool:
    return 'bool_value: %s' % ('true' if v else 'false')
  if typ is int:
    return 'int64_value: %d' % v
  if typ is float:
    return 'float_value: %s' % _float_text(v)
  return 'bytes_value: "%s"' % str(v).replace('\\', '\\\\').replace('"', '\\"')


def _float_text(v: float) -> str:
  # protobuf's text_format prints a float field as str() of the Python float with the shortest float32
  # round-trip digits (type_checkers.ToShortestFloat): '0.0001', '1e-05', '0.99' -- numpy's own str()
  # switches to the exponent form one decade earlier ('1e-04'), found by the google.protobuf cross-check
  # (tests/test_hparams_independent.py)
  return repr(float(str(np.float32(v))))


class HParamDefText:
  """str(obj) is the text-format of the HParamDef proto of `hparams` (map entries
  sorted by key, as protobuf prints maps)."""

  def __init__(self, hparams: HParams):
    self._hp = hparams

  def __str__(self):
    out = []
    for name in sorted(self._hp._hparam_types):
      typ, is_list = self._hp._hparam_types[name]
      v = getattr(self._hp, name)
      out.append('hparam {\n  key: "%s"\n  value {' % name)
      if is_list:
        kind = {bool: 'bool_list', int: 'int64_list', float: 'float_list'}.get(typ, 'bytes_list')
        out.append('    %s {' % kind)
        for item in v:
          if typ is bool:
            out.append('      value: %s' % ('true' if item else 'false'))
          elif typ is int:
            out.append('      value: %d' % item)
          elif typ is float:
            out.append('      value: %s' % _float_text(item))
          else:
            out.append('      value: "%s"' % item)
        out.append('    }')
      else:
        out.append('    ' + _proto_scalar(typ, v))
      out.append('  }\n}')
    return '\n'.join(out) + '\n'


def _parse_pbtxt(text: str) -> Dict[str, Any]:
  """Parses the HParamDef text proto into {name: python value}."""
  result = {}
  entry = re.compile(r'hparam\s*\{(.*?)\n\}', re.S)
  for m in entry.finditer(text):
    body = m.group(1)
    key = re.search(r'key:\s*"([^"]*)"', body).group(1)
    lst = re.search(r'(int64_list|float_list|bytes_list|bool_list)\s*\{(.*?)\}', body, re.S)
    if lst:
      kind, inner = lst.group(1), lst.group(2)
      vals = re.findall(r'value:\s*("(?:[^"\\]|\\.)*"|\S+)', inner)
      conv = {'int64_list': int, 'float_list': float,
              'bool_list': lambda s: s == 'true',
              'bytes_list': lambda s: s[1:-1]}[kind]
      result[key] = [conv(v) for v in vals]
      continue
    sc = re.search(r'(int64_value|float_value|bytes_value|bool_value):\s*("(?:[^"\\]|\\.)*"|\S+)',
                   body)
    if not sc:
      raise ValueError('cannot parse hparam entry for %s' % key)
    kind, raw = sc.group(1), sc.group(2)
    if kind == 'int64_value':
      result[key] = int(raw)
    elif kind == 'float_value':
      result[key] = float(raw)
    elif kind == 'bool_value':
      result[key] = raw == 'true'
    else:
      result[key] = raw[1:-1].replace('\\"', '"').replace('\\\\', '\\')
  return result


def create_hparams(**kwargs: Any) -> HParams:
  """Default hyper-parameters, names and values of utils.py:87-148."""
  hparams = HParams(
      checkpoint_dir='',
      supervisor_dir='',
      basis_file_path='',
      wavefunction_type='',
      composite_wavefunction_types=('', ''),
      wavefunction_optimizer_type='',
      num_sites=40,
      size_x=1,
      size_y=1,
      size_z=1,
      num_fc_layers=3,
      fc_layer_size=80,
      num_conv_layers=5,
      conv_strides=1,
      kernel_size=5,
      num_conv_filters=16,
      num_resnet_blocks=2,
      bond_dimension=4,
      top_lin_table_file='',
      bot_lin_table_file='',
      ed_vector_file='',
      adjacency_list_path='',
      nonlinearity='relu',
      output_activation='exp',
      composite_output_activations=('', ''),
      num_equilibration_sweeps=100,
      num_monte_carlo_sweeps=1,
      num_epochs=500,
      batch_size=200,
      num_batches_per_epoch=50,
      time_evolution_beta=0.12,
      learning_rates=[1e-3, 1e-4, 2e-5, 1e-5],
      learning_rate_stops=[300, 600, 1000],
      optimizer='adam',
      beta2=0.99,
      num_evaluation_samples=100,
      # extension (not in the reference): StochasticReconfiguration optimizer, training.py here
      sr_diag_shift=0.01,
      sr_cg_tolerance=1e-3,
      sr_cg_max_iterations=100,
  )
  hparams.override_from_dict(kwargs)
  return hparams


def load_hparams(hparams_path: str) -> HParams:
  """Reads the `hparams.pbtxt` text proto (utils.py:153-166)."""
  with open(hparams_path, 'r') as f:
    values = _parse_pbtxt(f.read())
  hparams = HParams()
  for k, v in values.items():
    hparams.add_hparam(k, v)
  return hparams


def random_configurations(n_sites: int, batch_size: int = 1, seed=None) -> np.ndarray:
  """Random Sz = 0 chains (utils.py:169-192): every row starts all-up and gets n_sites // 2
  distinct sites lowered; a candidate site that is already down is redrawn.  The draws come
  from RandomState(seed).randint in the reference's order (unseeded, like the reference, when
  `seed` is None), so a seeded call reproduces the reference's chains."""
  draw = np.random.RandomState(seed).randint
  n_down = n_sites // 2
  chains = np.ones((batch_size, n_sites), np.float32)
  for row in chains:
    site, lowered = draw(0, n_sites), 0
    while lowered < n_down:
      if row[site] > 0:
        row[site] = -1.0
        lowered += 1
      else:
        site = draw(0, n_sites)
  return chains
