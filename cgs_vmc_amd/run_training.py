"""Runs ground-state wavefunction optimization (counterpart of cgs_vmc/run_training.py).

Same flags, files and call order: hparams.pbtxt, optional J.txt bond list (else a 1-D periodic
chain), a checkpoint `model_prior_{epoch}_epochs` BEFORE every epoch and one energy per line
appended to metrics.txt (run_training.py:84-153).

  python -m cgs_vmc_amd.run_training --checkpoint_dir=/tmp/run --num_sites=16 \
      --wavefunction_type=fully_connected --optimizer=EnergyGradient --num_epochs=10 \
      --hparams="batch_size=64,fc_layer_size=32,num_fc_layers=2"
"""
from __future__ import annotations

import argparse
import os
import sys

from . import evaluation  # noqa: F401  (kept for parity with the reference's imports)
from . import lattice
from . import operators
from . import parallel
from . import session as session_lib
from . import training
from . import utils
from . import wavefunctions


def _bool(v):
  return str(v).lower() in ('1', 'true', 'yes')


def build_parser():
  p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawTextHelpFormatter)
  p.add_argument('--checkpoint_dir', default='', help='Full path to the checkpoint directory.')
  p.add_argument('--num_sites', type=int, default=24, help='Number of sites in the system.')
  p.add_argument('--heisenberg_jx', type=float, default=1.0, help='Jx value in Heisenberg Hamiltonian.')
  p.add_argument('--num_epochs', type=int, default=1000, help='Total of number of epochs to train on.')
  p.add_argument('--checkpoint_frequency', type=int, default=1, help='(unused, as in the reference)')
  p.add_argument('--resume_training', type=_bool, nargs='?', const=True, default=False)
  p.add_argument('--wavefunction_type', default='')
  p.add_argument('--optimizer', default='ITSWO')
  p.add_argument('--generate_vectors', type=_bool, nargs='?', const=True, default=False)
  p.add_argument('--basis_file_path', default='')
  p.add_argument('--hparams', default='')
  p.add_argument('--override', type=_bool, nargs='?', const=True, default=True)
  return p


def main(argv=None):
  FLAGS = build_parser().parse_args(argv)
  parallel.init_from_env('nccl')
  n_sites = FLAGS.num_sites
  hparams = utils.create_hparams()
  hparams.set_hparam('checkpoint_dir', FLAGS.checkpoint_dir)
  hparams.set_hparam('basis_file_path', FLAGS.basis_file_path)
  hparams.set_hparam('num_sites', FLAGS.num_sites)
  hparams.set_hparam('num_epochs', FLAGS.num_epochs)
  hparams.set_hparam('wavefunction_type', FLAGS.wavefunction_type)
  hparams.set_hparam('wavefunction_optimizer_type', FLAGS.optimizer)
  hparams.parse(FLAGS.hparams)
  hparams_path = os.path.join(hparams.checkpoint_dir, 'hparams.pbtxt')
  is_chief = parallel.rank() == 0

  if not os.path.exists(FLAGS.checkpoint_dir):
    os.makedirs(FLAGS.checkpoint_dir, exist_ok=True)

  if os.path.exists(hparams_path) and not FLAGS.override:
    print('Hparams file already exists')
    sys.exit()

  if is_chief:
    with open(hparams_path, 'w') as file:
      file.write(str(hparams.to_proto()))

  heisenberg_jx = FLAGS.heisenberg_jx
  heisenberg_bonds = lattice.load_bonds(FLAGS.checkpoint_dir, n_sites)

  wavefunction = wavefunctions.build_wavefunction(hparams)
  hamiltonian = operators.HeisenbergHamiltonian(heisenberg_bonds, heisenberg_jx, 1.)

  wavefunction_optimizer = training.GROUND_STATE_OPTIMIZERS[FLAGS.optimizer]()

  shared_resources = {}

  graph_building_args = {
      'wavefunction': wavefunction,
      'hamiltonian': hamiltonian,
      'hparams': hparams,
      'shared_resources': shared_resources
  }

  train_ops = wavefunction_optimizer.build_opt_ops(**graph_building_args)

  session = session_lib.Session()
  init = session_lib.global_variables_initializer()
  init_l = session_lib.local_variables_initializer()
  session.run([init, init_l])
  if parallel.world_size() > 1:
    # all ranks must start from identical parameters: rank 0's initial values win
    theta = wavefunction._get_theta()
    theta = parallel.allreduce_array(theta if is_chief else 0 * theta).astype('float32')
    wavefunction._set_theta(theta)

  checkpoint_saver = session_lib.Saver(wavefunction.get_trainable_variables(), max_to_keep=5)

  if FLAGS.resume_training:
    latest_checkpoint = session_lib.latest_checkpoint(hparams.checkpoint_dir)
    checkpoint_saver.restore(session, latest_checkpoint)

  training_metrics_file = os.path.join(hparams.checkpoint_dir, 'metrics.txt')
  for epoch_number in range(FLAGS.num_epochs):
    checkpoint_name = 'model_prior_{}_epochs'.format(epoch_number)
    save_path = os.path.join(hparams.checkpoint_dir, checkpoint_name)
    if is_chief:
      checkpoint_saver.save(session, save_path)

    metrics_record = wavefunction_optimizer.run_optimization_epoch(
        train_ops, session, hparams)

    if is_chief:
      metrics_file_output = open(training_metrics_file, 'a')
      metrics_file_output.write('{}\n'.format(metrics_record))
      metrics_file_output.close()

  if FLAGS.generate_vectors:
    raise NotImplementedError('--generate_vectors (VectorWavefunctionEvaluator) is outside '
                              'the MI355X hot path')


if __name__ == '__main__':
  main()
