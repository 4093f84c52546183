"""Ground-state optimisation driver; command-line counterpart of cgs_vmc/run_training.py.

Contract kept from the reference (run_training.py:21-153): the flags below, `hparams.pbtxt`
written into --checkpoint_dir, an optional `J.txt` bond list (else a periodic chain), one
checkpoint `model_prior_{epoch}_epochs` BEFORE every epoch (5 kept) and one energy per line
appended to `metrics.txt`.

  python -m cgs_vmc_amd.run_training --checkpoint_dir=/tmp/run --num_sites=16 \\
      --wavefunction_type=fully_connected --optimizer=EnergyGradient --num_epochs=10 \\
      --heisenberg_jx=-1.0 --hparams="batch_size=64,fc_layer_size=32,num_fc_layers=2"
"""
from __future__ import annotations

import os
import sys

from . import cli_common
from . import parallel
from . import session as session_lib
from . import training
from . import utils

FLAG_TABLE = (
    ('checkpoint_dir', str, '', 'Full path to the checkpoint directory.'),
    ('num_sites', int, 24, 'Number of sites in the system.'),
    ('heisenberg_jx', float, 1.0, 'Jx value in Heisenberg Hamiltonian.'),
    ('num_epochs', int, 1000, 'Total of number of epochs to train on.'),
    ('checkpoint_frequency', int, 1, 'Accepted and ignored, as in the reference.'),
    ('resume_training', bool, False, 'Restore variables from the latest checkpoint.'),
    ('wavefunction_type', str, '', 'Key of wavefunctions.WAVEFUNCTION_TYPES.'),
    ('optimizer', str, 'ITSWO', 'Key of training.GROUND_STATE_OPTIMIZERS.'),
    ('generate_vectors', bool, False, 'Not available on the MI355X path.'),
    ('basis_file_path', str, '', 'Basis file for --generate_vectors.'),
    ('hparams', str, '', 'Comma-separated name=value overrides of the hyper-parameters.'),
    ('override', bool, True, 'Overwrite an existing hparams.pbtxt.'),
)


def hparams_from_flags(flags):
  """create_hparams() + the flag-controlled fields + --hparams overrides (run_training.py:84-90)."""
  hp = utils.create_hparams()
  for field, value in (('checkpoint_dir', flags.checkpoint_dir),
                       ('basis_file_path', flags.basis_file_path),
                       ('num_sites', flags.num_sites),
                       ('num_epochs', flags.num_epochs),
                       ('wavefunction_type', flags.wavefunction_type),
                       ('wavefunction_optimizer_type', flags.optimizer)):
    hp.set_hparam(field, value)
  return hp.parse(flags.hparams)


def train(flags, hp):
  chief = parallel.rank() == 0
  run_dir = hp.checkpoint_dir
  ansatz, hamiltonian = cli_common.heisenberg_system(hp, flags.checkpoint_dir, flags.heisenberg_jx)
  optimizer = training.GROUND_STATE_OPTIMIZERS[flags.optimizer]()
  train_ops = optimizer.build_opt_ops(**cli_common.graph_kwargs(
      wavefunction=ansatz, hamiltonian=hamiltonian, hparams=hp))

  sess = session_lib.Session()
  sess.run([session_lib.global_variables_initializer(),
            session_lib.local_variables_initializer()])
  cli_common.broadcast_parameters(ansatz)

  saver = session_lib.Saver(ansatz.get_trainable_variables(), max_to_keep=5)
  if flags.resume_training:
    saver.restore(sess, session_lib.latest_checkpoint(run_dir))

  metrics_path = os.path.join(run_dir, 'metrics.txt')
  for epoch in range(flags.num_epochs):
    if chief:   # the checkpoint holds the parameters PRIOR to this epoch
      saver.save(sess, os.path.join(run_dir, 'model_prior_{}_epochs'.format(epoch)))
    energy = optimizer.run_optimization_epoch(train_ops, sess, hp)
    if chief:
      with open(metrics_path, 'a') as out:
        out.write('{}\n'.format(energy))
  return sess


def main(argv=None):
  flags = cli_common.parser_from_table(__doc__, FLAG_TABLE).parse_args(argv)
  parallel.init_from_env('nccl')
  hp = hparams_from_flags(flags)
  cli_common.ensure_directory(flags.checkpoint_dir)
  pbtxt = os.path.join(hp.checkpoint_dir, 'hparams.pbtxt')
  # rank 0 alone looks at the file system and every rank follows its verdict: a slower rank
  # must not mistake the file rank 0 is about to write for a pre-existing one
  exists = os.path.exists(pbtxt) if parallel.rank() == 0 else False
  if parallel.allreduce_max(1.0 if exists else 0.0) > 0.5 and not flags.override:
    print('Hparams file already exists')
    if parallel.is_distributed():
      parallel._dist().destroy_process_group()
    sys.exit()
  if parallel.rank() == 0:
    with open(pbtxt, 'w') as out:
      out.write(str(hp.to_proto()))
  train(flags, hp)
  if flags.generate_vectors:
    raise NotImplementedError('--generate_vectors (VectorWavefunctionEvaluator) is outside '
                              'the MI355X hot path')


if __name__ == '__main__':
  main()
