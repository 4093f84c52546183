"""Per-phase s_memtime ticks (shader clocks on gfx950) of one mc_step of the eight-chain sampler k_sweep8 (its stamped instantiation):
  python tools/prof_sweep8.py [N H L chains]     default: config 5's shard, then config 2"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cgs_vmc_amd.engine import VmcEngine  # noqa: E402


def run(n, h, L, b):
  theta, cfg = bench.make_inputs(n, h, L, b, 0)
  eng = VmcEngine(n, b, L, h)
  eng.set_params(theta); eng.set_configs(cfg)
  eng.sweep_tile(8)
  eng.mc_steps(n)
  ph = eng.debug_sweep_profile(2 * n)
  eng.close()
  tot = sum(ph.values())
  print('N={} H={} L={} chains={}: {:.0f} shader clocks per mc_step (= {:.2f} us at 2.4 GHz)'.format(n, h, L, b, tot, tot / 2400.0))
  for k, v in ph.items():
    if v:
      print('  {:<14s}{:8.1f}  {:5.1f} %'.format(k, v, 100.0 * v / tot))


if __name__ == '__main__':
  if len(sys.argv) >= 5:
    run(*[int(x) for x in sys.argv[1:5]])
  else:
    run(256, 256, 6, 1024)
    run(36, 128, 3, 1024)
