"""Is a bench step bound by the host's enqueue rate?  python tools/step_host.py [workload] [steps]
Runs `steps` steps of bench.py's single-GPU structure (reset + accumulate + one sweep) and prints the host time
spent enqueueing them (the loop alone, no synchronisation) next to the time until the GPU has finished them."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cgs_vmc_amd import _hip  # noqa: E402
from cgs_vmc_amd.engine import VmcEngine  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else 'heisenberg6x6_fc3x128_b1024'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
lx, ly, nnn, L, h, b = bench.WORKLOADS[wl][:6]
ansatz, ksz = (bench.WORKLOADS[wl][6:] + ('fully_connected', 0))[:2]
conv = ansatz in ('conv_2d', 'res_net_2d')
n = lx * ly
theta, cfg = bench.make_inputs(n, h, L, b, 0, ansatz, ksz)
eng = VmcEngine(n, b, L, h, seed=2024, ansatz=ansatz, kernel_size=ksz, size_x=ly if conv else 0, size_y=lx if conv else 0)
bonds = bench.torus_bonds(lx, ly, nnn)
eng.set_params(theta); eng.set_configs(cfg); eng.set_bonds(bonds, *bench.couplings(len(bonds), nnn))
for _ in range(int(os.environ.get('WARM_SWEEPS', '10'))):
  eng.mc_steps(n, want_accepted=False)


def step():
  eng.reset_accumulators()
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  eng.mc_steps(n, want_accepted=False)


def epoch(k):
  eng.epoch_energy_gradient(0, k, n, 1e10)


for name, fn in (('three calls per step', lambda: [step() for _ in range(steps)]),
                 ('one library call for all steps', lambda: epoch(steps))):
  for _ in range(int(os.environ.get('WARM_STEPS', '50'))):
    step()
  eng.synchronize()
  for rep in range(3):
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter()
    eng.synchronize()
    t2 = time.perf_counter()
    print('{} [{}]: host enqueue {:.1f} us per step, until the GPU is done {:.1f} us per step'.format(
        wl, name, 1e6 * (t1 - t0) / steps, 1e6 * (t2 - t0) / steps), flush=True)
eng.close()
