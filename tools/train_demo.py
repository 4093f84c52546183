import sys, os, time; sys.path.insert(0, '/root/repo')
os.environ['CGS_VMC_INIT_SEED']='3'; os.environ['CGS_VMC_CONFIG_SEED']='4'
import numpy as np, tempfile
from cgs_vmc_amd import lattice, run_training, run_energy_evaluation
d = tempfile.mkdtemp()
lattice.write_bonds(d, lattice.torus_bonds(4, 4))
hp = ('batch_size=512,fc_layer_size=64,num_fc_layers=2,num_equilibration_sweeps=10,'
      'num_batches_per_epoch=20,learning_rates=[0.003,0.001,0.0003],learning_rate_stops=[150,300],'
      'num_evaluation_samples=50')
t0=time.time()
run_training.main(['--checkpoint_dir', d, '--num_sites', '16', '--heisenberg_jx', '-1.0',
                   '--wavefunction_type', 'fully_connected', '--optimizer', sys.argv[1] if len(sys.argv)>1 else 'EnergyGradient',
                   '--num_epochs', '400', '--hparams', hp])
m = [float(x) for x in open(os.path.join(d, 'metrics.txt')).read().split()]
print('train time', time.time()-t0, 'energies', m[0], m[50], m[100], m[200], m[300], m[-1], 'exact -11.2285')
