"""Energy evaluation driver; command-line counterpart of cgs_vmc/run_energy_evaluation.py.

Reads `hparams.pbtxt` (+ optional `J.txt`) and the latest checkpoint of --checkpoint_dir, runs
MonteCarloOperatorEvaluator and prints `Energy: <mean> +/- <uncertainty>`.  The uncertainty
uses the reference's expression sqrt(std)/n (run_energy_evaluation.py:87, defect B6 kept so the
line is comparable); the conventional standard error follows on a second line.
"""
from __future__ import annotations

import os

import numpy as np

from . import cli_common
from . import evaluation
from . import parallel
from . import session as session_lib
from . import utils

FLAG_TABLE = (
    ('heisenberg_jx', float, 1.0, 'Jx value in Heisenberg Hamiltonian.'),
    ('checkpoint_dir', str, '', 'Full path to the checkpoint directory.'),
    ('output_file', str, '', 'Accepted and unused, as in the reference.'),
    ('hparams', str, '', 'Comma-separated name=value overrides of the hyper-parameters.'),
)


def evaluate(flags):
  hp = utils.load_hparams(os.path.join(flags.checkpoint_dir, 'hparams.pbtxt'))
  hp.parse(flags.hparams)
  ansatz, hamiltonian = cli_common.heisenberg_system(hp, flags.checkpoint_dir, flags.heisenberg_jx)
  evaluator = evaluation.MonteCarloOperatorEvaluator()
  eval_ops = evaluator.build_eval_ops(**cli_common.graph_kwargs(
      wavefunction=ansatz, operator=hamiltonian, hparams=hp))
  sess = session_lib.Session()
  sess.run(session_lib.global_variables_initializer())
  session_lib.Saver(ansatz.get_trainable_variables()).restore(
      sess, session_lib.latest_checkpoint(hp.checkpoint_dir))
  return np.asarray(evaluator.run_evaluation(eval_ops, sess, hp, epoch_num=0), np.float64)


def main(argv=None):
  flags = cli_common.parser_from_table(__doc__, FLAG_TABLE).parse_args(argv)
  parallel.init_from_env('nccl')
  samples = evaluate(flags)
  mean_energy = samples.mean()
  uncertainty = np.sqrt(samples.std()) / samples.size
  if parallel.rank() == 0:
    print('Energy: {} +/- {}'.format(mean_energy, uncertainty))
    print('Standard error (std/sqrt(n)): {}'.format(samples.std() / np.sqrt(samples.size)))
  return mean_energy, uncertainty


if __name__ == '__main__':
  main()
