"""Executor shim: the small part of the TensorFlow-1 graph/session surface the reference's
drivers use (SURVEY.md 8b "Executor").

An op handle is any object with `_run()`; `Session.run(handle_or_list)` executes handles in
order, synchronously, and returns their values (side-effecting ops return None or a value
callers ignore; `metrics` / `energy` / `value` return a Python float), as
training.py:608-623, 750-763 and evaluation.py:138-145 expect.
"""
from __future__ import annotations

import os
import re
from typing import Any, Callable, List, Optional, Sequence

import numpy as np


class Op:
  """A runnable handle.  `fn` is executed on every Session.run."""

  def __init__(self, fn: Callable[[], Any], name: str = 'op'):
    self._fn = fn
    self.name = name

  def _run(self):
    return self._fn()

  def __repr__(self):
    return '<Op %s>' % self.name


class Tensor(Op):
  """An Op whose value is an array / scalar (e.g. psi, local energy, mean energy)."""


def group(*ops) -> Op:
  """tf.group: runs all inputs, returns None."""
  flat = []
  for o in ops:
    flat.extend(o if isinstance(o, (list, tuple)) else [o])

  def run():
    for o in flat:
      if o is not None:
        o._run()
  return Op(run, 'group')


class Graph:
  """Registry of stateful objects for the initializer ops (tf default graph stand-in)."""

  def __init__(self):
    self.global_initializers: List[Callable[[], None]] = []
    self.local_initializers: List[Callable[[], None]] = []
    self.named = {}

  def reset(self):
    self.__init__()


_default_graph = Graph()


def get_default_graph() -> Graph:
  return _default_graph


def reset_default_graph():
  _default_graph.reset()


def global_variables_initializer() -> Op:
  """Initialises trainable variables, chains, epoch counter, Adam slots."""
  g = _default_graph
  return Op(lambda: [f() for f in list(g.global_initializers)] and None, 'init')


def local_variables_initializer() -> Op:
  """tf.local_variables_initializer: the metric accumulators."""
  g = _default_graph
  return Op(lambda: [f() for f in list(g.local_initializers)] and None, 'init_local')


class Session:
  """Synchronous executor of op handles (tf.Session stand-in)."""

  def run(self, fetches):
    if fetches is None:
      return None
    if isinstance(fetches, (list, tuple)):
      return [self.run(f) for f in fetches]
    if isinstance(fetches, dict):
      return {k: self.run(v) for k, v in fetches.items()}
    if not hasattr(fetches, '_run'):
      raise TypeError('Fetch argument %r is not an op handle' % (fetches,))
    return fetches._run()

  def close(self):
    pass

  def __enter__(self):
    return self

  def __exit__(self, *exc):
    self.close()


# --------------------------------------------------------------------------- #
# Checkpoints: tf.train.Saver stand-in (run_training.py:134-146,
# run_energy_evaluation.py:80-83).  Format: <prefix>.npz keyed by variable name, plus the
# `checkpoint` state file with TensorFlow's text layout so latest_checkpoint works the same.
# --------------------------------------------------------------------------- #
class Variable:
  """A named view on (part of) an engine-resident or host array."""

  def __init__(self, name: str, shape: Sequence[int], getter: Callable[[], np.ndarray],
               setter: Callable[[np.ndarray], None], trainable: bool = True):
    self.name = name
    self.shape = tuple(shape)
    self._getter = getter
    self._setter = setter
    self.trainable = trainable

  def eval(self) -> np.ndarray:
    return np.asarray(self._getter()).reshape(self.shape)

  def load(self, value):
    value = np.asarray(value, np.float32)
    if value.shape != self.shape:
      raise ValueError('shape mismatch for %s: %s vs %s' % (self.name, value.shape, self.shape))
    self._setter(value)

  def __repr__(self):
    return '<Variable %s %s>' % (self.name, self.shape)


class Saver:
  def __init__(self, var_list: Sequence[Variable], max_to_keep: Optional[int] = 5):
    self._vars = list(var_list)
    self._max_to_keep = max_to_keep
    self._kept: List[str] = []

  def save(self, session: Session, save_path: str) -> str:
    """Writes `<save_path>.npz` keyed by the TF variable names, or -- with
    CGS_VMC_CHECKPOINT_FORMAT=tf -- a TensorFlow V2 bundle (`.index` + `.data-00000-of-00001`,
    cgs_vmc_amd/tf_checkpoint.py) that the reference's tf.train.Saver can restore."""
    del session
    arrays = {v.name: v.eval() for v in self._vars}
    if os.environ.get('CGS_VMC_CHECKPOINT_FORMAT', 'npz') == 'tf':
      from . import tf_checkpoint
      tf_checkpoint.write_bundle(save_path, arrays)
    else:
      np.savez(save_path + '.npz', **arrays)
    self._kept.append(save_path)
    if self._max_to_keep and len(self._kept) > self._max_to_keep:
      old = self._kept.pop(0)
      for suffix in ('.npz', '.index', '.data-00000-of-00001'):
        if os.path.exists(old + suffix):
          os.remove(old + suffix)
    directory = os.path.dirname(save_path)
    with open(os.path.join(directory, 'checkpoint'), 'w') as f:
      f.write('model_checkpoint_path: "%s"\n' % os.path.basename(save_path))
      for p in self._kept:
        f.write('all_model_checkpoint_paths: "%s"\n' % os.path.basename(p))
    return save_path

  def restore(self, session: Session, save_path: str):
    del session
    if save_path is None:
      raise ValueError("Can't load save_path when it is None.")
    if os.path.exists(save_path + '.npz'):
      data = np.load(save_path + '.npz')
    else:   # a checkpoint written by the reference's tf.train.Saver (run_training.py:134-146)
      from . import tf_checkpoint
      if not tf_checkpoint.bundle_exists(save_path):
        raise FileNotFoundError('no checkpoint %s(.npz | .index)' % save_path)
      data = tf_checkpoint.read_bundle(save_path)
    for v in self._vars:
      if v.name not in data:
        raise KeyError('variable %s not found in checkpoint %s' % (v.name, save_path))
      v.load(data[v.name])


def latest_checkpoint(checkpoint_dir: str) -> Optional[str]:
  """tf.train.latest_checkpoint: reads the `checkpoint` state file."""
  state = os.path.join(checkpoint_dir, 'checkpoint')
  if not os.path.exists(state):
    return None
  with open(state) as f:
    m = re.search(r'model_checkpoint_path:\s*"([^"]*)"', f.read())
  if not m:
    return None
  path = m.group(1)
  if not os.path.isabs(path):
    path = os.path.join(checkpoint_dir, path)
  return path if (os.path.exists(path + '.npz') or os.path.exists(path + '.index')) else None
