# compare the CU-sharing pair (CGS_VMC_CO=1) with the separate kernels (CGS_VMC_CO=0): chains bit for
# bit, accumulators within float tolerance, and time per step
import os, sys, time, subprocess, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def run(co):
  os.environ['CGS_VMC_CO'] = str(co)
  import torch
  import bench
  from cgs_vmc_amd import _hip
  from cgs_vmc_amd.engine import VmcEngine
  lx, ly, nnn, L, h, b = bench.WORKLOADS['heisenberg10x10_fc3x256_b4096'][:6]
  n = lx * ly
  bonds = bench.torus_bonds(lx, ly, nnn); jx, jz = bench.couplings(len(bonds), nnn)
  theta, cfg = bench.make_inputs(n, h, L, b, 0)
  eng = VmcEngine(n, b, L, h, device=0, chain_offset=0, seed=2024)
  eng.set_params(theta); eng.set_configs(cfg); eng.set_bonds(bonds, jx, jz)
  eng.mc_steps(n, want_accepted=False)
  accs = []
  for i in range(4):
    eng.reset_accumulators()
    eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
    eng.mc_steps(n, want_accepted=False)
    accs.append(eng.get_accumulators().copy())
  cfgs = eng.get_configs().copy()
  eng.synchronize()
  def step():
    eng.reset_accumulators(); eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT); eng.mc_steps(n, want_accepted=False)
  for _ in range(10): step()
  eng.synchronize(); torch.cuda.synchronize()
  ts = []
  for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(50): step()
    eng.synchronize()
    ts.append((time.perf_counter() - t0) / 50 * 1e3)
  np.savez('gpurun_out/co2/run%d.npz' % co, accs=np.array(accs), cfgs=cfgs, ms=np.array(ts))
  print('CO=%d ms/step' % co, ['%.4f' % t for t in ts], flush=True)

if __name__ == '__main__':
  if len(sys.argv) > 1:
    run(int(sys.argv[1]))
  else:
    os.makedirs('gpurun_out/co2', exist_ok=True)
    for co in (0, 1):
      subprocess.run([sys.executable, __file__, str(co)], check=True, timeout=300)
    a = np.load('gpurun_out/co2/run0.npz'); b = np.load('gpurun_out/co2/run1.npz')
    print('chains identical:', np.array_equal(a['cfgs'], b['cfgs']))
    for i in range(len(a['accs'])):
      x, y = a['accs'][i], b['accs'][i]
      print('acc', i, 'max abs diff', float(np.abs(x - y).max()), 'scale', float(np.abs(x).max()),
            'rel', float(np.abs(x - y).max() / np.abs(x).max()))
    print('median ms: CO=0 %.4f  CO=1 %.4f' % (np.median(a['ms']), np.median(b['ms'])))
