"""Operator interface and the Heisenberg Hamiltonian (mirror of cgs_vmc/operators.py).

The arithmetic of HeisenbergHamiltonian.build / local_value / apply_in_place runs in
libcgsvmc_hip.so: k_bond_count / k_bond_fill list the antiparallel bonds of every chain,
k_tail16 evaluates psi(swap_ij R)/psi(R) for exactly those rows and k_eloc_reduce sums them
per chain.  The results equal the reference's "evaluate every bond, then mask"
(operators.py:166-168) because masked rows contribute exactly zero.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from . import session as session_lib


class Operator():
  """Operators base class (operators.py:13-87)."""

  def build(self, wavefunction, inputs, psi=None):
    raise NotImplementedError

  def local_value(self, wavefunction, inputs, psi=None):
    raise NotImplementedError

  def apply_in_place(self, wavefunction, inputs, psi=None):
    raise NotImplementedError

  def apply(self, wavefunction):
    raise NotImplementedError


class HeisenbergHamiltonian(Operator):
  """sum over bonds of S_i.S_j with couplings j_x (transverse) and j_z
  (operators.py:212-287).  `j_x` / `j_z` may also be per-bond sequences (extension; the
  reference takes one value for all bonds, operators.py:215-225)."""

  def __init__(self, bonds: List[Tuple[int, int]], j_x, j_z):
    self._bonds_list = [(int(b[0]), int(b[1])) for b in bonds]
    nb = len(self._bonds_list)
    self._j_x = np.broadcast_to(np.asarray(j_x, np.float32), (nb,)).copy()
    self._j_z = np.broadcast_to(np.asarray(j_z, np.float32), (nb,)).copy()

  def _engine_for(self, wavefunction, inputs):
    from . import graph_builders
    if not isinstance(inputs, graph_builders.ConfigsVariable):
      raise TypeError('the MI355X Hamiltonian kernels act on the CONFIGS variable '
                      '(graph_builders.get_configs)')
    engine = wavefunction._bind(inputs)
    inputs._ensure_hamiltonian(self)
    return engine

  def build(self, wavefunction, inputs, psi=None) -> Tuple[session_lib.Tensor, ...]:
    """(diagonal matrix element, off-diagonal term sum_bonds 0.5 jx [s_i s_j<0] psi(R_ij)),
    operators.py:227-247."""
    del psi
    engine = self._engine_for(wavefunction, inputs)

    def diag():
      inputs._ensure_hamiltonian(self)      # another operator may have been evaluated in between
      return engine.local_energy_terms(wavefunction._which)[0]

    def offdiag():
      inputs._ensure_hamiltonian(self)
      d, o = engine.local_energy_terms(wavefunction._which)
      return o * engine.amplitude(None, wavefunction._which)[1]
    return session_lib.Tensor(diag, 'sz_elements'), session_lib.Tensor(offdiag, 's_perp_terms')

  def local_value(self, wavefunction, inputs, psi=None) -> session_lib.Tensor:
    """diag + s_perp / psi (operators.py:249-259)."""
    del psi
    engine = self._engine_for(wavefunction, inputs)
    return LocalValueTensor(self, wavefunction, inputs, engine)

  def apply_in_place(self, wavefunction, inputs, psi=None) -> session_lib.Tensor:
    """diag * psi + s_perp (operators.py:261-271)."""
    del psi
    engine = self._engine_for(wavefunction, inputs)

    def value():
      inputs._ensure_hamiltonian(self)
      eloc, _ = engine.local_energy(wavefunction._which)
      return eloc * engine.amplitude(None, wavefunction._which)[1]
    return session_lib.Tensor(value, 'h_psi')

  def apply(self, wavefunction):
    raise NotImplementedError('TransformedWavefunction (operators.py:90-125) is used by no '
                              'driver and is outside the MI355X hot path')


class HeisenbergBond(HeisenbergHamiltonian):
  """S_i . S_j on one bond (operators.py:128-169): `build` returns
  (0.25 j_z s_i s_j, 0.25 j_x 2 [s_i s_j < 0] psi(R with i and j exchanged)).  It is the
  one-bond case of the Hamiltonian kernels (the connected-row list then has at most one row per
  chain)."""

  def __init__(self, bond: Tuple[int, int], j_x, j_z):
    super(HeisenbergBond, self).__init__([bond], j_x, j_z)
    self._bond = (int(bond[0]), int(bond[1]))


class LocalValueTensor(session_lib.Tensor):
  """E_loc[B] of `operator` under `wavefunction` on the chains."""

  def __init__(self, operator, wavefunction, configs, engine):
    self.operator, self.wavefunction, self.configs, self.engine = (
        operator, wavefunction, configs, engine)
    super(LocalValueTensor, self).__init__(self._value, 'local_value')

  def _value(self):
    self.configs._ensure_hamiltonian(self.operator)
    return self.engine.local_energy(self.wavefunction._which)[0]

  def mean(self) -> float:
    """Batch mean over ALL ranks' chains (evaluation.py:102 tf.reduce_mean)."""
    from . import parallel
    self.configs._ensure_hamiltonian(self.operator)
    _, m = self.engine.local_energy(self.wavefunction._which, want_eloc=False)
    if parallel.world_size() > 1:
      m = parallel.allreduce_sum(m) / parallel.world_size()
    return m


def reduce_mean(tensor) -> session_lib.Tensor:
  """tf.reduce_mean for the tensors of this module."""
  if isinstance(tensor, LocalValueTensor):
    out = session_lib.Tensor(lambda: np.float32(tensor.mean()), 'mean')
    out.local_value_tensor = tensor       # evaluation.run_evaluation fuses the loop over it
    return out
  return session_lib.Tensor(lambda: np.float32(np.mean(tensor._run())), 'mean')
