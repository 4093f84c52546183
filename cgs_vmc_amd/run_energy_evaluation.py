"""Runs evaluation on the optimized wave-function (counterpart of
cgs_vmc/run_energy_evaluation.py): loads hparams.pbtxt and the latest checkpoint from
--checkpoint_dir, runs MonteCarloOperatorEvaluator and prints `Energy: mean +/- uncertainty`
with the reference's formula (sqrt(std)/n, run_energy_evaluation.py:86-88; defect B6 kept
for output parity, the standard error std/sqrt(n) is printed on a second line)."""
from __future__ import annotations

import argparse
import os

import numpy as np

from . import evaluation
from . import lattice
from . import operators
from . import parallel
from . import session as session_lib
from . import utils
from . import wavefunctions


def build_parser():
  p = argparse.ArgumentParser(description=__doc__)
  p.add_argument('--heisenberg_jx', type=float, default=1.0)
  p.add_argument('--checkpoint_dir', default='')
  p.add_argument('--output_file', default='')
  p.add_argument('--hparams', default='')
  return p


def main(argv=None):
  """Evaluates energy and prints the result."""
  FLAGS = build_parser().parse_args(argv)
  parallel.init_from_env('nccl')
  hparams_path = os.path.join(FLAGS.checkpoint_dir, 'hparams.pbtxt')
  hparams = utils.load_hparams(hparams_path)
  hparams.parse(FLAGS.hparams)  # optional way to override some hparameters
  n_sites = hparams.num_sites

  heisenberg_jx = FLAGS.heisenberg_jx
  heisenberg_bonds = lattice.load_bonds(FLAGS.checkpoint_dir, n_sites)

  wavefunction = wavefunctions.build_wavefunction(hparams)
  hamiltonian = operators.HeisenbergHamiltonian(heisenberg_bonds, heisenberg_jx, 1.)

  evaluator = evaluation.MonteCarloOperatorEvaluator()

  shared_resources = {}

  graph_building_args = {
      'wavefunction': wavefunction,
      'operator': hamiltonian,
      'hparams': hparams,
      'shared_resources': shared_resources
  }

  evaluation_ops = evaluator.build_eval_ops(**graph_building_args)

  init = session_lib.global_variables_initializer()
  session = session_lib.Session()
  session.run(init)

  checkpoint_manager = session_lib.Saver(wavefunction.get_trainable_variables())

  latest_checkpoint = session_lib.latest_checkpoint(hparams.checkpoint_dir)
  checkpoint_manager.restore(session, latest_checkpoint)

  data = evaluator.run_evaluation(evaluation_ops, session, hparams, epoch_num=0)
  mean_energy = np.mean(data)
  uncertainty = np.sqrt(np.std(data)) / len(data)
  if parallel.rank() == 0:
    print('Energy: {} +/- {}'.format(mean_energy, uncertainty))
    print('Standard error (std/sqrt(n)): {}'.format(np.std(data) / np.sqrt(len(data))))
  return mean_energy, uncertainty


if __name__ == '__main__':
  main()
