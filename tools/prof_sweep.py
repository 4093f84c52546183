"""Per-phase shader cycles of one mc_step of the sampler (s_memtime-stamped diagnostic build) at
config 3, for all waves and separately for the chain-owning waves 0-3 and the others."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cgs_vmc_amd.engine import VmcEngine
n, h, L, b = 100, 256, 3, 4096
theta, cfg = bench.make_inputs(n, h, L, b, 0)
eng = VmcEngine(n, b, L, h); eng.set_params(theta); eng.set_configs(cfg)
eng.set_bonds(bench.torus_bonds(10, 10, False), -1.0, 1.0)
eng.mc_steps(200)
cols = {}
for label, mask in (('all', '0xff'), ('waves 0-3', '0x0f'), ('waves 4-7', '0xf0')):
  os.environ['CGS_VMC_PROFILE_WAVES'] = mask
  cols[label] = eng.debug_sweep_profile(200)
print('%-18s' % 'phase' + ''.join('%12s' % k for k in cols))
for k in cols['all']:
  print('%-18s' % k + ''.join('%12.0f' % cols[c][k] for c in cols))
print('%-18s' % 'total' + ''.join('%12.0f' % sum(cols[c].values()) for c in cols))
