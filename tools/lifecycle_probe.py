"""Diagnostic for VERDICT r2 item 6: "No HIP GPUs are available" from torch's lazy init after ~140
engine life cycles.  Creates and destroys engines of mixed ansatz types BEFORE torch touches the
GPU and prints, every few cycles, what a leak would show up in: open file descriptors, mappings,
resident memory, threads and the device's free memory (hipMemGetInfo through libamdhip64).
Usage: python tools/lifecycle_probe.py [cycles] [--torch-first]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def proc_stats():
  fds = len(os.listdir('/proc/self/fd'))
  with open('/proc/self/maps') as f:
    maps = sum(1 for _ in f)
  rss = thr = 0
  with open('/proc/self/status') as f:
    for line in f:
      if line.startswith('VmRSS:'):
        rss = int(line.split()[1]) // 1024
      if line.startswith('Threads:'):
        thr = int(line.split()[1])
  return fds, maps, rss, thr


def fd_kinds():
  kinds = {}
  for name in os.listdir('/proc/self/fd'):
    try:
      t = os.readlink('/proc/self/fd/' + name)
    except OSError:
      continue
    key = t.split(':')[0] if not t.startswith('/') else t
    kinds[key] = kinds.get(key, 0) + 1
  return kinds


def main():
  cycles = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 300
  if '--torch-first' in sys.argv:
    import torch
    torch.cuda.init()
  from cgs_vmc_amd.engine import VmcEngine
  hip = C.CDLL('libamdhip64.so')
  free, total = C.c_size_t(), C.c_size_t()

  def dev_free():
    hip.hipMemGetInfo(C.byref(free), C.byref(total))
    return free.value >> 20

  specs = [dict(n_sites=16, batch_size=64, num_layers=2, layer_size=32),
           dict(n_sites=36, batch_size=128, num_layers=3, layer_size=128),
           dict(n_sites=16, batch_size=64, num_layers=1, layer_size=32, ansatz='rbm'),
           dict(n_sites=16, batch_size=64, num_layers=2, layer_size=8, ansatz='conv_2d', kernel_size=3, size_x=4, size_y=4),
           dict(n_sites=16, batch_size=64, num_layers=2, layer_size=300),
           # (round 5) beyond the fused convolution kernels: the general path of conv_general.hip, ~1 GB of block buffers per ctx
           dict(n_sites=16, batch_size=64, num_layers=2, layer_size=72, ansatz='conv_2d', kernel_size=3, size_x=4, size_y=4)]
  bonds = [(i, (i + 1) % 16) for i in range(16)]
  print('cycle fds maps rss_MB threads dev_free_MB')
  seen_maps, last_rss = None, 0
  for k in range(cycles):
    spec = specs[k % len(specs)]
    eng = VmcEngine(seed=k, **spec)
    rng = np.random.default_rng(k)
    eng.set_params((0.1 * rng.standard_normal(eng.num_params)).astype(np.float32))
    n, b = spec['n_sites'], spec['batch_size']
    cfg = np.ones((b, n), np.float32); cfg[:, ::2] = -1.0
    eng.set_configs(cfg)
    eng.set_bonds(bonds if n == 16 else [(i, (i + 1) % n) for i in range(n)], -1.0, 1.0)
    eng.mc_steps(n)
    eng.reset_accumulators(); eng.accumulate(0)
    eng.amplitude(cfg[:7])
    eng.close()
    if k % 20 == 0 or k == cycles - 1:
      st = proc_stats()
      print(k, *st, dev_free(), flush=True)
      with open('/proc/self/maps') as f:
        now = set(line.strip() for line in f)
      if seen_maps is not None and st[2] - last_rss > 50:      # a step in resident memory: which mappings are new
        for line in sorted(now - seen_maps):
          lo, hi = (int(x, 16) for x in line.split()[0].split('-'))
          print('  new mapping {:8.1f} MB  {}'.format((hi - lo) / 2**20, ' '.join(line.split()[1:])), flush=True)
      seen_maps, last_rss = now, st[2]
  print('fd kinds:', fd_kinds())
  import torch
  try:
    torch.cuda.init()
    x = torch.ones(4, device='cuda')
    print('torch after {} cycles: ok'.format(cycles), float(x.sum()))
  except Exception as e:  # pylint: disable=broad-except
    print('torch after {} cycles: FAILED {}'.format(cycles, e))
    print(proc_stats(), fd_kinds())
    sys.exit(1)


if __name__ == '__main__':
  main()
