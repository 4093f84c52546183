"""End-to-end physics check on the GPU box: train the dense ansatz with the EnergyGradient optimizer through the
reference-shaped CLI (cgs_vmc_amd.run_training) and compare the variational energy with the known ground-state
energy of the periodic spin-1/2 Heisenberg antiferromagnet -- an answer no part of this repository computed:
  4 x 4 torus:   E0 = -11.228483 (exact diagonalisation, also tests/exact_states.py)
  6 x 6 torus:   E0 / N = -0.678872 (exact diagonalisation, Schulz, Ziman, Poilblanc 1996)
  10 x 10 torus: E0 / N = -0.671549(4) (quantum Monte Carlo, Sandvik 1997)
The variational principle bounds every epoch energy from below by E0 (up to the Monte Carlo error); a wrong
sampler distribution, local energy, gradient estimator or optimizer shows up as an energy that falls through the
bound or does not approach it.  Prints one JSON line per lattice.  usage: python tools/train_demo.py [small]"""
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = [
    # name, lx, ly, wavefunction_type, optimizer, ansatz hparams, chains, epochs, learning rates, exact energy per site
    ('4x4', 4, 4, 'fully_connected', 'EnergyGradient', 'fc_layer_size=64,num_fc_layers=2', 512, 200,
     (0.001, 0.0003, 0.0001), -11.228483 / 16),
    ('4x4', 4, 4, 'fully_connected', 'StochasticReconfiguration',
     'fc_layer_size=64,num_fc_layers=2,sr_diag_shift=0.01,sr_cg_tolerance=0.001,sr_cg_max_iterations=200', 512, 100,
     (0.05, 0.02, 0.01), -11.228483 / 16),
    ('6x6', 6, 6, 'fully_connected', 'EnergyGradient', 'fc_layer_size=128,num_fc_layers=3', 1024, 300,
     (0.001, 0.0003, 0.0001), -0.678872),
    ('6x6', 6, 6, 'conv_2d', 'EnergyGradient', 'size_x=6,size_y=6,num_conv_layers=4,num_conv_filters=16,kernel_size=3',
     1024, 300, (0.001, 0.0003, 0.0001), -0.678872),
    ('10x10', 10, 10, 'fully_connected', 'EnergyGradient', 'fc_layer_size=256,num_fc_layers=3', 4096, 250,
     (0.001, 0.0003, 0.0001), -0.671549),
]


def main():
  from cgs_vmc_amd import lattice, run_energy_evaluation, run_training, session, wavefunctions
  cases = CASES[:4] if len(sys.argv) > 1 and sys.argv[1] == 'small' else CASES
  for name, lx, ly, wf_type, optimizer, ansatz_hp, b, epochs, lrs, exact in cases:
    n = lx * ly
    d = tempfile.mkdtemp(prefix='cgsvmc_demo_')
    try:
      lattice.write_bonds(d, lattice.torus_bonds(lx, ly))
      session.reset_default_graph(); wavefunctions.reset_name_scope()
      hp = ('batch_size={},{},num_equilibration_sweeps=20,num_batches_per_epoch=50,'
            'learning_rates=[{},{},{}],learning_rate_stops=[{},{}],num_evaluation_samples=50').format(
                b, ansatz_hp, lrs[0], lrs[1], lrs[2], epochs // 2, (3 * epochs) // 4)
      t0 = time.time()
      run_training.main(['--checkpoint_dir', d, '--num_sites', str(n), '--heisenberg_jx', '-1.0',
                         '--wavefunction_type', wf_type, '--optimizer', optimizer,
                         '--num_epochs', str(epochs), '--hparams', hp])
      t_train = time.time() - t0
      metrics = [float(x) for x in open(os.path.join(d, 'metrics.txt')).read().split()]
      session.reset_default_graph(); wavefunctions.reset_name_scope()
      mean, unc = run_energy_evaluation.main(['--checkpoint_dir', d, '--heisenberg_jx', '-1.0'])
      tail = metrics[-10:]
      out = {
          'lattice': name + ' torus', 'wavefunction_type': wf_type, 'optimizer': optimizer, 'hparams': ansatz_hp,
          'chains': b, 'epochs': epochs, 'batches_per_epoch': 50, 'train_seconds': round(t_train, 1),
          'energy_per_site_first_epoch': metrics[0] / n,
          'energy_per_site_last_10_epochs_mean': sum(tail) / len(tail) / n,
          'energy_per_site_min_epoch': min(metrics) / n,
          'evaluation_energy_per_site': mean / n, 'evaluation_uncertainty_per_site': unc / n,
          'exact_energy_per_site': exact,
          'relative_error_of_evaluation': (mean / n - exact) / abs(exact),
          'below_exact_by_more_than_5_sigma': bool(mean / n < exact - 5 * unc / n),
      }
      print(json.dumps(out), flush=True)
    finally:
      shutil.rmtree(d, ignore_errors=True)


if __name__ == '__main__':
  main()
