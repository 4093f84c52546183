"""Times the fused dense sampler ALONE (no accumulate beside it) with sixteen- and eight-chain tiles:
  python tools/sweep_tiles.py [workload] [chains ...]
Prints ms per sweep launch (num_sites mc_steps, HIP events on the library's stream) and us per mc_step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def run(workload, b, tile, reps=8):
  from cgs_vmc_amd.engine import VmcEngine
  if workload.startswith('shape:'):        # shape:N,H,L
    n, h, L = [int(x) for x in workload[6:].split(',')]
  else:
    lx, ly, nnn, L, h = bench.WORKLOADS[workload][:5]
    n = lx * ly
  theta, cfg = bench.make_inputs(n, h, L, min(b, 4096), 0)
  cfg = cfg[[i % cfg.shape[0] for i in range(b)]]
  eng = VmcEngine(n, b, L, h, seed=2024)
  eng.set_params(theta)
  eng.set_configs(cfg)
  if eng.sweep_tile(tile) != tile:
    raise RuntimeError('tile not available')
  for _ in range(3):
    eng.mc_steps(n, want_accepted=False)
  eng.timing_enable(1)
  eng.timing_reset()
  for _ in range(reps):
    eng.mc_steps(n, want_accepted=False)
  eng.synchronize()
  ms, cnt = eng.timing_get('sweep')
  eng.close()
  return ms / cnt, n


if __name__ == '__main__':
  wl = sys.argv[1] if len(sys.argv) > 1 else 'heisenberg16x16j1j2_fc6x256_b1024'
  chains = [int(x) for x in sys.argv[2:]] or [bench.WORKLOADS[wl][5]]
  tiles = [int(x) for x in os.environ.get('TILES', '16,8').split(',')]
  for b in chains:
    for tile in tiles:
      try:
        ms, n = run(wl, b, tile)
        print('{} chains={} tile={}: {:.4f} ms/sweep = {:.2f} us per mc_step'.format(wl, b, tile, ms, ms * 1e3 / n), flush=True)
      except Exception as e:  # pylint: disable=broad-except
        print('{} chains={} tile={}: {!r}'.format(wl, b, tile, e), flush=True)
