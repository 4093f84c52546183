"""Local energies only (no sampler) at a bench workload, for rocprofv3 --kernel-trace --stats of the row path alone:
  python tools/eloc_only.py [workload] [calls]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cgs_vmc_amd.engine import VmcEngine  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else 'heisenberg36x36_conv3x16k5_b32'
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 5
w = bench.WORKLOADS[wl]
lx, ly, nnn, L, h, b = w[:6]
ansatz, ksz = (w[6:] + ('fully_connected', 0))[:2]
n = lx * ly
theta, cfg = bench.make_inputs(n, h, L, b, 0, ansatz, ksz)
kw = dict(ansatz=ansatz, kernel_size=ksz, size_x=lx, size_y=ly) if ansatz != 'fully_connected' else {}
eng = VmcEngine(n, b, L, h, **kw)
eng.set_params(theta); eng.set_configs(cfg); eng.set_bonds(bench.torus_bonds(lx, ly, nnn), -1.0, 1.0)
eng.local_energy(want_eloc=False)
eng.timing_enable(1); eng.timing_reset()
for _ in range(calls):
  eng.set_configs(cfg)
  eng.local_energy(want_eloc=False)
eng.synchronize()
ms, cnt = eng.timing_get('tail_eloc')
print('{}: {:.3f} ms per local-energy call ({} rows)'.format(wl, ms / max(cnt, 1), eng.last_connected_rows()))
eng.close()
