"""The general convolution sampler with and without the patch sampler (csrc/conv_patch.hip), sweep time by batch:
  python tools/conv_patch_bench.py [lx ly layers filters kernel] [chains ...]
Prints ms per sweep (n_sites mc_steps, HIP events on the library's stream) for CGS_VMC_CONV_PATCH=0 (a full forward of
every candidate per step) and =1 (the two boxes per convolution that the exchanged pair reaches), same seeds: the chains
are the same chains (tests/test_gpu_conv_general.py), so the accept counts printed beside them must agree."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cgs_vmc_amd.engine import VmcEngine  # noqa: E402


def run(lx, ly, L, f, k, b, patch, sweeps):
  os.environ['CGS_VMC_CONV_PATCH'] = patch
  n = lx * ly
  theta, cfg = bench.make_inputs(n, f, L, b, 0, 'conv_2d', k)
  eng = VmcEngine(n, b, L, f, seed=2024, ansatz='conv_2d', kernel_size=k, size_x=ly, size_y=lx)
  eng.set_params(theta)
  eng.set_configs(cfg)
  eng.mc_steps(n, want_accepted=False)
  eng.timing_enable(1)
  eng.timing_reset()
  acc = 0
  for _ in range(sweeps):
    acc += eng.mc_steps(n)
  eng.synchronize()
  ms, cnt = eng.timing_get('sweep')
  used = eng.conv_patch(n)
  eng.close()
  return ms / cnt, acc, used


if __name__ == '__main__':
  args = [int(x) for x in sys.argv[1:]]
  lx, ly, L, f, k = args[:5] if len(args) >= 5 else (36, 36, 3, 16, 5)
  chains = args[5:] or [32, 256]
  for b in chains:
    sweeps = 2 if b <= 64 else 1
    full, acc0, _ = run(lx, ly, L, f, k, b, '0', sweeps)
    patch, acc1, used = run(lx, ly, L, f, k, b, '1', sweeps)
    print('{}x{} conv_2d {} x {} filters {}x{}, {} chains: full forward {:.2f} ms per sweep, patch sampler{} {:.2f} ms ({:.1f} x); '
          'accepted {} / {}'.format(lx, ly, L, f, k, k, b, full, '' if used else ' (NOT taken)', patch, full / patch, acc0, acc1), flush=True)
