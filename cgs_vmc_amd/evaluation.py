"""Evaluation of optimized wavefunctions (mirror of cgs_vmc/evaluation.py, hot-path scope:
MonteCarloOperatorEvaluator)."""
from __future__ import annotations

import os
from typing import Any, Dict, List, NamedTuple

import numpy as np

from . import graph_builders
from . import operators
from . import parallel
from . import session as session_lib
from .training import _run_mc_steps

EvalOps = NamedTuple(
    'EvaluationOps', [
        ('value', session_lib.Op),
        ('mc_step', session_lib.Op),
        ('acceptance_rate', session_lib.Op),
        ('placeholder_input', session_lib.Op),
        ('wavefunction_value', session_lib.Op),
    ]
)
"""Named tuple of tensors representing evaluation components."""


class WavefunctionEvaluator():
  """Parents class for wavefunction evaluators (evaluation.py:27-71)."""

  def build_eval_ops(self, wavefunction, operator, hparams, shared_resources):
    raise NotImplementedError

  def run_evaluation(self, eval_ops, session, hparams, epoch_num: int) -> Any:
    raise NotImplementedError


def _fused_evaluation(eval_ops):
  """(engine, ensure_hamiltonian) when `eval_ops` are the handles build_eval_ops made -- the batch
  mean of a Hamiltonian's local value under the wavefunction that also drives mc_step on the same
  CONFIGS variable -- else None (the op-by-op loop serves anything else).  CGS_VMC_EVAL_FUSED=0
  forces the op-by-op loop."""
  if os.environ.get('CGS_VMC_EVAL_FUSED', '1') == '0':
    return None
  lv = getattr(eval_ops.value, 'local_value_tensor', None)
  mc = eval_ops.mc_step
  if not isinstance(lv, operators.LocalValueTensor) or not hasattr(mc, 'run_many'):
    return None
  if lv.configs is not getattr(mc, 'configs', None) or lv.wavefunction is not getattr(mc, 'wavefunction', None):
    return None
  if lv.wavefunction._which != 0 or not hasattr(lv.engine, 'evaluate'):
    return None
  return lv.engine, lambda: lv.configs._ensure_hamiltonian(lv.operator)


class MonteCarloOperatorEvaluator(WavefunctionEvaluator):
  """Operator evaluation by running MCMC (evaluation.py:74-152)."""

  def build_eval_ops(self, wavefunction, operator, hparams,
                     shared_resources: Dict[graph_builders.ResourceName, Any]) -> EvalOps:
    """evaluation.py:77-110."""
    batch_size = hparams.batch_size
    n_sites = hparams.num_sites

    configs = graph_builders.get_configs(shared_resources, batch_size, n_sites)
    mc_step, acc_rate = graph_builders.get_monte_carlo_sampling(
        shared_resources, configs, wavefunction)

    value = operators.reduce_mean(operator.local_value(wavefunction, configs))
    eval_ops = EvalOps(
        value=value,
        mc_step=mc_step,
        acceptance_rate=acc_rate,
        placeholder_input=None,
        wavefunction_value=None,
    )
    return eval_ops

  def run_evaluation(self, eval_ops: EvalOps, session, hparams, epoch_num: int) -> List[float]:
    """evaluation.py:113-152: thermalise for num_equilibration_sweeps sweeps, then take
    num_evaluation_samples measurements of the batch-mean local value, num_monte_carlo_sweeps
    sweeps apart.  Each block of num_sites consecutive mc_steps is one persistent-kernel
    launch; the acceptance count the reference computes and drops is kept in
    `self.acceptance_count`."""
    del epoch_num
    steps_per_sweep = hparams.num_sites
    decorrelation = hparams.num_monte_carlo_sweeps * steps_per_sweep
    self.acceptance_count = 0
    fused = _fused_evaluation(eval_ops)
    if fused is not None:
      # the whole loop in ONE host call (vmc_evaluate): batch sums stay on the device, sharded
      # chains are reduced in one float64 all-reduce at the end -- the same list of float32 means
      engine, ensure = fused
      ensure()
      coll = parallel.collective() if parallel.world_size() > 1 else None
      means, accepted = engine.evaluate(coll, hparams.num_equilibration_sweeps * steps_per_sweep,
                                        hparams.num_evaluation_samples, decorrelation)
      self.acceptance_count = accepted
      eval_ops.mc_step.last_accepted = accepted
      return [np.float32(m) for m in means]

    def measurements():
      _run_mc_steps(session, eval_ops.mc_step, hparams.num_equilibration_sweeps * steps_per_sweep)
      for _ in range(hparams.num_evaluation_samples):
        yield session.run(eval_ops.value)
        _run_mc_steps(session, eval_ops.mc_step, decorrelation)
        self.acceptance_count += getattr(eval_ops.mc_step, 'last_accepted', 0)

    return list(measurements())


class VectorWavefunctionEvaluator(WavefunctionEvaluator):
  """evaluation.py:155-246: dumps psi over a basis file.  Offline tool outside the hot path
  (SURVEY.md 2); `Wavefunction.__call__` on an array gives the same amplitudes."""

  def build_eval_ops(self, wavefunction, operator, hparams, shared_resources):
    raise NotImplementedError('VectorWavefunctionEvaluator is outside the MI355X hot path')
