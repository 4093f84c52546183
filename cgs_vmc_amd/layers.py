"""Activation registry (mirror of cgs_vmc/layers.py:13-21).

In the reference the values are TensorFlow functions; here they are named tokens that
the engine maps to kernel epilogues (cgsvmc.h VMC_ACT_*): all seven are available as the hidden
nonlinearity (template parameter of the fused row / sampler kernels; relu is the tuned path) and
as the output activation (exp works in the log domain with exp_norm_shift, any other g gives
psi = g(x) and linear-domain ratios).  The periodic convolutions and residual blocks of layers.py
(Conv1dPeriodic / Conv2dPeriodic / ResBlock1d / ResBlock2d, layers.py:24-229) have no Python class here: they
exist as kernels (csrc/conv_kernels.hpp) behind wavefunctions.Conv1DNetwork / Conv2DNetwork / ResNet1D / ResNet2D.
The MPS and graph-convolution building blocks serve ansaetze outside the hot path and are
not provided.
"""
import numpy as np


class Nonlinearity:
  """A named activation; callable on numpy arrays for host-side use."""

  def __init__(self, name, fn):
    self.name = name
    self._fn = fn

  def __call__(self, x):
    return self._fn(np.asarray(x))

  def __repr__(self):
    return 'Nonlinearity(%s)' % self.name

  def __eq__(self, other):
    return isinstance(other, Nonlinearity) and other.name == self.name

  def __hash__(self):
    return hash(self.name)


NONLINEARITIES = {
    'relu': Nonlinearity('relu', lambda x: np.maximum(x, 0)),
    'exp': Nonlinearity('exp', np.exp),
    'cos': Nonlinearity('cos', np.cos),
    'tan': Nonlinearity('tan', np.tan),
    'tanh': Nonlinearity('tanh', np.tanh),
    'sigmoid': Nonlinearity('sigmoid', lambda x: 1.0 / (1.0 + np.exp(-x))),
    'identity': Nonlinearity('identity', lambda x: x),
}
