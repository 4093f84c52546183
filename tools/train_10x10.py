"""End-to-end run at BASELINE config 3: 10x10 Heisenberg torus (Marshall-rotated, j_x = -1),
fully-connected 3x256 ansatz, 4096 chains, through the run_training counterpart.
Reference value: E0/N = -0.6715 (QMC, Sandvik) for the 10x10 periodic lattice.
Usage: python tools/train_10x10.py [optimizer] [epochs] [wavefunction_type] [lr0,lr1]
wavefunction_type conv_2d / res_net_2d: the hparams defaults of utils.py:108-114 (5 layers or 2
blocks of 16 filters, 5x5 kernels) on the same lattice."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('CGS_VMC_INIT_SEED', '3')
os.environ.setdefault('CGS_VMC_CONFIG_SEED', '4')
from cgs_vmc_amd import lattice, run_training  # noqa: E402

opt = sys.argv[1] if len(sys.argv) > 1 else 'EnergyGradient'
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 200
wf_type = sys.argv[3] if len(sys.argv) > 3 else 'fully_connected'
d = tempfile.mkdtemp()
lattice.write_bonds(d, lattice.torus_bonds(10, 10))
hp = ('batch_size=4096,fc_layer_size=256,num_fc_layers=3,num_equilibration_sweeps=10,size_x=10,size_y=10,'
      'num_batches_per_epoch=10,learning_rates=[0.001,0.0003],learning_rate_stops=[150]')
if len(sys.argv) > 4:       # learning rates, e.g. "0.0002,0.0001"
  hp = hp.replace('learning_rates=[0.001,0.0003]', 'learning_rates=[%s]' % sys.argv[4])
if opt == 'StochasticReconfiguration':
  hp = hp.replace('learning_rates=[0.001,0.0003]', 'learning_rates=[0.03,0.01]')
t0 = time.time()
run_training.main(['--checkpoint_dir', d, '--num_sites', '100', '--heisenberg_jx', '-1.0',
                   '--wavefunction_type', wf_type, '--optimizer', opt,
                   '--num_epochs', str(epochs), '--checkpoint_frequency', '1000', '--hparams', hp])
m = [float(x) / 100 for x in open(os.path.join(d, 'metrics.txt')).read().split()]
dt = time.time() - t0
print('wavefunction_type', wf_type, 'optimizer', opt, 'epochs', epochs, 'wall s', round(dt, 1), 's/epoch', round(dt / epochs, 3))
print('E/N every %d epochs:' % max(1, epochs // 10), [round(x, 4) for x in m[::max(1, epochs // 10)]], 'last', round(m[-1], 4),
      '(QMC -0.6715)')
