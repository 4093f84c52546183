"""The patch kernels (csrc/conv_patch.hip) against full forwards over a long run:
  python tools/conv_patch_soak.py [sweeps] [lx ly layers filters kernel chains]
Two engines from the same start (36 x 36 sites, 3 x 16 filters 5 x 5, 32 chains by default), one with the patch kernels
for sampler and local energies, one with a full forward of every candidate and connected configuration
(CGS_VMC_CONV_PATCH=0, read per call).  After every sweep: chains, cached logits, accept count and local energies must be
the same bits.  Prints the first difference, or the totals."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cgs_vmc_amd.engine import VmcEngine  # noqa: E402

sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
lx, ly, L, f, k, b = [int(x) for x in sys.argv[2:8]] if len(sys.argv) >= 8 else (36, 36, 3, 16, 5, 32)
n = lx * ly
theta, cfg = bench.make_inputs(n, f, L, b, 0, 'conv_2d', k)
bonds = bench.torus_bonds(lx, ly, False)
engines = []
for _ in range(2):
  e = VmcEngine(n, b, L, f, seed=2024, ansatz='conv_2d', kernel_size=k, size_x=ly, size_y=lx)
  e.set_params(theta); e.set_configs(cfg); e.set_bonds(bonds, *bench.couplings(len(bonds), False))
  engines.append(e)
assert engines[0].kernel_path() == 6 and engines[0].conv_patch(n)
t = [0.0, 0.0]
accepted = [0, 0]
energy = 0.0
for s in range(sweeps):
  out = []
  for i, mode in enumerate(('1', '0')):
    os.environ['CGS_VMC_CONV_PATCH'] = mode
    t0 = time.perf_counter()
    acc = engines[i].mc_steps(n)
    eloc, mean = engines[i].local_energy()
    t[i] += time.perf_counter() - t0
    accepted[i] += acc
    out.append((acc, engines[i].get_configs(), engines[i].amplitude()[0], eloc))
  for name, x, y in zip(('accept count', 'chains', 'logits', 'local energies'), out[0], out[1]):
    if not np.array_equal(x, y):
      print('sweep {}: {} differ'.format(s, name))
      sys.exit(1)
  energy = float(np.mean(out[0][3])) / n
print('{} x {} conv_2d {} x {} filters {}x{}, {} chains: {} sweeps = {} steps per chain + {} local-energy calls; chains, logits, accept '
      'counts ({} of {}) and local energies bit-identical after every sweep; energy per site {:.5f}; wall {:.1f} s on the patch '
      'kernels, {:.1f} s on full forwards'.format(lx, ly, L, f, k, k, b, sweeps, sweeps * n, sweeps, accepted[0], sweeps * n * b, energy,
                                                  t[0], t[1]))
for e in engines:
  e.close()
