import sys; sys.path.insert(0, '/root/repo')
import numpy as np, bench
from cgs_vmc_amd.engine import VmcEngine
n,h,L,b = 100,256,3,4096
theta,cfg = bench.make_inputs(n,h,L,b,0)
eng = VmcEngine(n,b,L,h); eng.set_params(theta); eng.set_configs(cfg); eng.set_bonds(bench.torus_bonds(10,10,False),-1.0,1.0)
eng.mc_steps(200)
p = eng.debug_sweep_profile(200)
tot = sum(p.values())
for k,v in p.items(): print('%-14s %9.0f cyc  %5.1f%%' % (k, v, 100*v/tot))
print('total', tot)
