"""Long-run check that the sampler tile never changes a result: N epochs of EnergyGradient training (device-resident epochs +
Adam) at BASELINE config 5's shard shape and at config 2, once on sixteen- and once on eight-chain sampler tiles from the
same start -- parameters, Adam moments, chains and epoch energies must be the same bits after every epoch.
  python tools/tile_soak.py [epochs]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cgs_vmc_amd.engine import VmcEngine  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for wl, batches in (('heisenberg16x16j1j2_fc6x256_b1024', 6), ('heisenberg6x6_fc3x128_b1024', 20)):
  lx, ly, nnn, L, h, b = bench.WORKLOADS[wl][:6]
  n = lx * ly
  theta, cfg = bench.make_inputs(n, h, L, b, 0)
  eng = VmcEngine(n, b, L, h, seed=2024)
  eng.set_bonds(bench.torus_bonds(lx, ly, nnn), -1.0, 1.0)
  runs = {}
  for tile in (16, 8):
    eng.sweep_tile(tile)
    eng.set_params(theta)
    eng.set_adam_state(np.zeros(theta.size, np.float32), np.zeros(theta.size, np.float32), 0)
    eng.set_shift(-10.0)
    eng.set_configs(cfg)
    eng.step_counter = 0
    eng.synchronize()
    t0 = time.perf_counter()
    hist = []
    for ep in range(epochs):
      eng.epoch_energy_gradient(5 * n, batches, n, 1e10)
      e = eng.apply_adam(0, 1e-3)
      hist.append((e, eng.get_params().copy(), eng.get_configs().copy()))
    eng.synchronize()
    runs[tile] = (hist, eng.get_adam_state(), time.perf_counter() - t0)
  same = all(a[0] == b2[0] and np.array_equal(a[1], b2[1]) and np.array_equal(a[2], b2[2])
             for a, b2 in zip(runs[16][0], runs[8][0]))
  same = same and all(np.array_equal(x, y) for x, y in zip(runs[16][1][:2], runs[8][1][:2]))
  moved = float(np.abs(runs[8][0][-1][1] - theta).max())
  print('{}: {} epochs x {} batches, {} sampler steps per tile run: bit-identical after every epoch: {}; max |theta - theta0| = {:.3f}; '
        'energy per site {:.4f} -> {:.4f}; wall {:.1f} s (sixteen-chain tiles) / {:.1f} s (eight-chain tiles)'.format(
            wl, epochs, batches, epochs * (5 * n + batches * n), 'YES' if same else 'NO', moved,
            runs[8][0][0][0] / n, runs[8][0][-1][0] / n, runs[16][2], runs[8][2]))
  eng.close()
  if not same:
    sys.exit(1)
