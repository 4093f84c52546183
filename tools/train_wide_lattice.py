"""End-to-end training on a lattice wider than the network's reach: 20 x 20 Heisenberg torus (Marshall-rotated, j_x = -1),
conv_2d 3 x 16 filters 3 x 3, 256 chains, through the run_training counterpart -- once on the fused kernels
(CGS_VMC_CONV_GENERAL=0) and once where plan_desc sends the shape by itself: the general path's patch kernels
(csrc/conv_patch.hip).  The two paths round differently, so the histories agree in trend, not in bits.  (40 epochs at 1e-4: the
plain-sum output of this ansatz with Adam leaves the stable range near epoch 45 on BOTH paths, -9.2 / -7.1 per site at epoch 50 --
the instability DESIGN_HISTORY.md H7 describes for the convolutional types at these hyper-parameters, not a path's.)
Usage: python tools/train_wide_lattice.py [epochs] [learning_rate]      (QMC: E0/N = -0.6699 on the 20 x 20 torus is the scale to read against)"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('CGS_VMC_INIT_SEED', '3')
os.environ.setdefault('CGS_VMC_CONFIG_SEED', '4')
from cgs_vmc_amd import lattice, run_training, session as session_lib, wavefunctions  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
lr = sys.argv[2] if len(sys.argv) > 2 else '0.0001'
hp = ('batch_size=256,num_conv_layers=3,num_conv_filters=16,kernel_size=3,num_equilibration_sweeps=5,size_x=20,size_y=20,'
      'num_batches_per_epoch=10,learning_rates=[%s],learning_rate_stops=[]' % lr)
for name, env in (('fused kernels (CGS_VMC_CONV_GENERAL=0)', '0'), ('routed to the patch kernels (default)', None)):
  if env is None:
    os.environ.pop('CGS_VMC_CONV_GENERAL', None)
  else:
    os.environ['CGS_VMC_CONV_GENERAL'] = env
  session_lib.reset_default_graph()
  wavefunctions.reset_name_scope()
  d = tempfile.mkdtemp()
  lattice.write_bonds(d, lattice.torus_bonds(20, 20))
  t0 = time.time()
  run_training.main(['--checkpoint_dir', d, '--num_sites', '400', '--heisenberg_jx', '-1.0', '--wavefunction_type', 'conv_2d',
                     '--optimizer', 'EnergyGradient', '--num_epochs', str(epochs), '--checkpoint_frequency', '1000', '--hparams', hp])
  dt = time.time() - t0
  m = [float(x) / 400 for x in open(os.path.join(d, 'metrics.txt')).read().split()]
  print('{}: {} epochs in {:.1f} s = {:.3f} s per epoch; E/N every {} epochs: {} last {:.4f}'.format(
      name, epochs, dt, dt / epochs, max(1, epochs // 6), [round(x, 4) for x in m[::max(1, epochs // 6)]], m[-1]), flush=True)
