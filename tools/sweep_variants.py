"""Times the sampler variants (waves per workgroup, W1 in LDS or L2) at a given shape:
  python tools/sweep_variants.py [workload] [chains ...]
Prints ms per sweep launch (num_sites mc_steps) from HIP events on the library's stream."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def run(workload, b, waves, w1l, reps=6):
  os.environ['CGS_VMC_SWEEP_WAVES'] = str(waves)
  os.environ['CGS_VMC_SWEEP_W1L'] = str(w1l)
  from cgs_vmc_amd.engine import VmcEngine
  lx, ly, nnn, L, h, _ = bench.WORKLOADS[workload]
  n = lx * ly
  theta, cfg = bench.make_inputs(n, h, L, min(b, 4096), 0)
  cfg = cfg[[i % cfg.shape[0] for i in range(b)]]
  eng = VmcEngine(n, b, L, h, seed=2024)
  eng.set_params(theta)
  eng.set_configs(cfg)
  for _ in range(3):
    eng.mc_steps(n, want_accepted=False)
  eng.timing_enable(1)
  eng.timing_reset()
  for _ in range(reps):
    eng.mc_steps(n, want_accepted=False)
  eng.synchronize()
  ms, cnt = eng.timing_get('sweep')
  eng.close()
  return ms / cnt


if __name__ == '__main__':
  wl = sys.argv[1] if len(sys.argv) > 1 else 'heisenberg10x10_fc3x256_b4096'
  chains = [int(x) for x in sys.argv[2:]] or [4096, 8192]
  for b in chains:
    for waves, w1l in ((8, 1), (8, 0), (4, 1), (4, 0)):
      try:
        print('{} chains={} waves={} w1_in_lds={}: {:.4f} ms/sweep'.format(wl, b, waves, w1l, run(wl, b, waves, w1l)), flush=True)
      except Exception as e:  # pylint: disable=broad-except
        print('{} chains={} waves={} w1_in_lds={}: {!r}'.format(wl, b, waves, w1l, e), flush=True)
