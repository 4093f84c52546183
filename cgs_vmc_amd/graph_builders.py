"""Graph-construction utilities (mirror of cgs_vmc/graph_builders.py).

`get_configs` creates the CONFIGS variable: the batch of Markov chains.  On the MI355X path
the variable lives in GPU memory inside a `VmcEngine`; with torch.distributed initialised
each rank owns batch_size / world_size consecutive chains and the Philox streams are keyed
by the GLOBAL chain id, so results do not depend on the number of GPUs.
"""
from __future__ import annotations

import enum
import os
from typing import Dict, Tuple

import numpy as np

from . import _hip
from . import parallel
from . import session as session_lib
from . import utils


class ResourceName(enum.Enum):
  """Type of sharable resources, serves as key in `shared_resources`
  (graph_builders.py:16-22)."""
  CONFIGS = 'CONFIGS'
  TARGET_CONFIGS = 'TARGET_CONFIGS'
  TARGET_PSI = 'TARGET_PSI'
  TRAINING_PSI = 'TRAINING_PSI'
  MONTE_CARLO_SAMPLING = 'MONTE_CARLO_SAMPLING'


class _Counter(session_lib.Tensor):
  """Global int32 `num_epochs` variable (graph_builders.py:25-35)."""

  def __init__(self):
    self.value = 0
    super(_Counter, self).__init__(lambda: self.value, 'num_epochs')
    session_lib.get_default_graph().global_initializers.append(self._init)

  def _init(self):
    self.value = 0


def get_or_create_num_epochs() -> _Counter:
  """Returns the variable counting optimisation epochs (tf.AUTO_REUSE semantics)."""
  g = session_lib.get_default_graph()
  if 'num_epochs' not in g.named:
    g.named['num_epochs'] = _Counter()
  return g.named['num_epochs']


def sampler_seed() -> int:
  """Key of the sampler's counter-based RNG for a new set of chains.  The reference draws
  from unseeded tf.random_uniform (graph_builders.py:59, 76), so repeated runs -- and
  repeated evaluations inside one process -- are independent samples: unless CGS_VMC_SEED
  pins it, every call returns fresh entropy drawn on rank 0 and shared with every rank (the
  chains of all ranks must use one key: it is combined with the global chain id).  The key is
  printed so that a run can be reproduced."""
  pinned = os.environ.get('CGS_VMC_SEED')
  if pinned is not None:
    return int(pinned)
  fresh = int.from_bytes(os.urandom(6), 'little')      # < 2^48: exact in float64
  if parallel.world_size() > 1:
    fresh = int(parallel.allreduce_array(
        np.array([float(fresh) if parallel.rank() == 0 else 0.0]))[0])
  if parallel.rank() == 0:
    print('sampler seed (CGS_VMC_SEED to reproduce): {}'.format(fresh), flush=True)
  return fresh


class ConfigsVariable:
  """Non-trainable variable of logical shape [batch_size, n_sites] holding the chains.

  `shape` is the global logical shape; this process owns `local_batch` rows starting at
  global chain `chain_offset`."""

  def __init__(self, name: str, batch_size: int, n_sites: int, seed=None):
    self.name = name
    self.shape = (batch_size, n_sites)
    self.local_batch, self.chain_offset = parallel.shard(batch_size)
    self._seed = seed
    # the sampler key is drawn (and shared over ranks: a collective) HERE, where every rank
    # constructs the variable, not at first engine use, which a rank may reach in another order
    self._sampler_seed = sampler_seed()
    self._engine = None
    self._engine_spec = None
    self._slots = {}
    self._hamiltonian = None
    self._host_value = utils.random_configurations(
        n_sites, self.local_batch, None if seed is None else seed + self.chain_offset)
    session_lib.get_default_graph().global_initializers.append(self._init)

  def _init(self):
    """tf.global_variables_initializer re-assigns the initial value."""
    if self._engine is not None:
      self._engine.set_configs(self._host_value)

  def get_shape(self):
    return self

  def as_list(self):
    return list(self.shape)

  # -- engine ---------------------------------------------------------------
  def _get_engine(self, wavefunction):
    if self._engine is None:
      from .engine import VmcEngine
      seed = self._sampler_seed
      self._engine = VmcEngine(
          n_sites=self.shape[1], batch_size=self.local_batch,
          device=parallel.local_rank(), chain_offset=self.chain_offset, seed=seed,
          **wavefunction._engine_spec())
      self._engine_spec = wavefunction._engine_spec()
      self._engine.set_configs(self._host_value)
    elif self._engine_spec != wavefunction._engine_spec():
      raise ValueError('a CONFIGS variable serves one ansatz shape (psi and its deep copy)')
    return self._engine

  def _claim_slot(self, wavefunction) -> int:
    for which in (_hip.VMC_PSI, _hip.VMC_OMEGA):
      if which not in self._slots:
        self._slots[which] = wavefunction
        return which
    raise ValueError('a CONFIGS variable supports two parameter sets: the wavefunction and '
                     'its supervisor copy')

  def _ensure_hamiltonian(self, hamiltonian):
    if self._hamiltonian is not hamiltonian:
      self._engine.set_bonds(hamiltonian._bonds_list, hamiltonian._j_x, hamiltonian._j_z)
      self._hamiltonian = hamiltonian

  # -- value ----------------------------------------------------------------
  def eval(self) -> np.ndarray:
    """This rank's chains [local_batch, n_sites]."""
    return self._engine.get_configs() if self._engine is not None else self._host_value.copy()

  def load(self, value):
    value = np.ascontiguousarray(value, np.float32)
    if value.shape != (self.local_batch, self.shape[1]):
      raise ValueError('Size of existing variable does not match.')
    self._host_value = value
    if self._engine is not None:
      self._engine.set_configs(value)

  def _run(self):
    return self.eval()


class _McStep(session_lib.Op):
  """`mc_step` handle.  Session.run(handle) performs one exchange step; the epoch loops of
  training / evaluation call `run_many(n)` to keep n steps in one persistent kernel launch
  (SURVEY.md 8f-1)."""

  def __init__(self, configs: ConfigsVariable, wavefunction):
    self.configs = configs
    self.wavefunction = wavefunction
    self.last_accepted = 0
    super(_McStep, self).__init__(lambda: self.run_many(1), 'mc_step')

  def run_many(self, n_steps: int):
    eng = self.configs._engine
    self.last_accepted = eng.mc_steps(n_steps)
    return None


def build_monte_carlo_sampling(inputs: ConfigsVariable, wavefunction, psi=None
                               ) -> Tuple[session_lib.Op, session_lib.Tensor]:
  """One exchange proposal + Metropolis accept per chain (graph_builders.py:38-89).

  Returns (mc_step, acceptance_count).  `psi` is accepted for signature parity; the kernel
  keeps the current amplitude of every chain cached on the GPU instead of recomputing it.
  """
  del psi
  if not isinstance(inputs, ConfigsVariable):
    raise TypeError('inputs must be the CONFIGS variable returned by get_configs')
  if wavefunction._bind(inputs) is None or wavefunction._which != _hip.VMC_PSI:
    raise ValueError('Monte-Carlo sampling must use the first wavefunction bound to CONFIGS')
  mc_step = _McStep(inputs, wavefunction)
  acceptance_count = session_lib.Tensor(lambda: np.float32(mc_step.last_accepted),
                                        'acceptance_count')
  return mc_step, acceptance_count


def get_configs(shared_resources: Dict[ResourceName, object], batch_size: int, n_sites: int,
                include: bool = True, configs_id: ResourceName = ResourceName.CONFIGS
                ) -> ConfigsVariable:
  """Retrieves or creates the variable holding a batch of configurations
  (graph_builders.py:92-125).

  Raises:
    ValueError: Size of existing variable does not match.
  """
  if configs_id in shared_resources:
    configs = shared_resources[configs_id]
    if configs.as_list() != [batch_size, n_sites]:
      raise ValueError('Size of existing variable does not match.')
    return configs
  seed = os.environ.get('CGS_VMC_CONFIG_SEED')
  configs = ConfigsVariable(str(configs_id), batch_size, n_sites,
                            None if seed is None else int(seed))
  if include:
    shared_resources[configs_id] = configs
  return configs


def get_monte_carlo_sampling(shared_resources: Dict[ResourceName, object],
                             inputs: ConfigsVariable, wavefunction, include: bool = True
                             ) -> Tuple[session_lib.Op, session_lib.Tensor]:
  """Memoised build_monte_carlo_sampling (graph_builders.py:128-151)."""
  if ResourceName.MONTE_CARLO_SAMPLING in shared_resources:
    return shared_resources[ResourceName.MONTE_CARLO_SAMPLING]
  mc_step, acc_rate = build_monte_carlo_sampling(inputs, wavefunction)
  if include:
    shared_resources[ResourceName.MONTE_CARLO_SAMPLING] = (mc_step, acc_rate)
  return mc_step, acc_rate
