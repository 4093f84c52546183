import sys, os; sys.path.insert(0, '/root/repo')
os.environ['CGS_VMC_SEED']='77'; os.environ['CGS_VMC_CONFIG_SEED']='5'; os.environ['CGS_VMC_INIT_SEED']='31'
import numpy as np
from oracle import vmc_oracle as vo
sys.path.insert(0, '/root/repo/tests')
import test_gpu_api as T
from cgs_vmc_amd import graph_builders, _hip
hp = T._hparams()
wf, ham, opt, ops, sess, shared = T._build('EnergyGradient', hp)
n, h, L, b = hp.num_sites, hp.fc_layer_size, hp.num_fc_layers, hp.batch_size
theta = wf._get_theta().copy()
cfg = shared[graph_builders.ResourceName.CONFIGS].eval()
eng = wf._engine
acc = vo.Accumulators(theta.size, np.float64)
vo.energy_gradient_accumulate(acc, theta, cfg, ham._bonds_list, -1.0, 1.0, -10.0, h, L, np.float64)
eng.reset_accumulators(); eng.accumulate(0)
g = eng.get_gradient(0); gref = vo.energy_gradient(acc)
a = eng.get_accumulators(); p = theta.size
d = np.abs(g-gref)
idx = np.argsort(-d)[:5]
names, shapes = wf._shapes()
offs = np.cumsum([0]+[int(np.prod(s)) for s in shapes])
for i in idx:
  k = np.searchsorted(offs, i, side='right')-1
  print(i, names[k], i-offs[k], 'gpu', g[i], 'ref', gref[i], 'g1', a[i], acc.g1_total[i], 'g2', a[p+i], acc.g2_total[i])
zero = np.where(gref == 0)[0]
print('exact-zero ref grads:', len(zero), 'gpu nonzero among them:', np.sum(g[zero] != 0), g[zero][g[zero]!=0][:5])
print('---- full epoch')
from cgs_vmc_amd import session, wavefunctions
session.reset_default_graph(); wavefunctions.reset_name_scope()
wf, ham, opt, ops, sess, shared = T._build('EnergyGradient', hp)
theta = wf._get_theta().copy()
cfg = shared[graph_builders.ResourceName.CONFIGS].eval()
bonds = ham._bonds_list
adam = vo.AdamState(theta.size); step = 0
cfg = T._oracle_sweeps(theta, cfg, hp.num_equilibration_sweeps * n, step, hp); step += hp.num_equilibration_sweeps * n
acc = vo.Accumulators(theta.size, np.float64)
for _ in range(hp.num_batches_per_epoch):
  vo.energy_gradient_accumulate(acc, theta, cfg, bonds, -1.0, 1.0, -10.0, h, L, np.float64)
  cfg = T._oracle_sweeps(theta, cfg, hp.num_monte_carlo_sweeps * n, step, hp); step += n
gref = vo.energy_gradient(acc)
theta_ref = vo.adam_apply(adam, theta, gref, 1e-2, 0.9, hp.beta2, 1e-8)
eng = wf._engine
# replicate epoch by hand on the engine to grab the gradient before Adam
eng.mc_steps(hp.num_equilibration_sweeps * n); eng.update_norm(1e10); eng.reset_accumulators()
for _ in range(hp.num_batches_per_epoch):
  eng.accumulate(0); eng.mc_steps(n)
g = eng.get_gradient(0)
eng.apply_adam(0, 1e-2, 0.9, hp.beta2, 1e-8)
th = eng.get_params()
d = np.abs(th - theta_ref); d[-1] = 0
for i in np.argsort(-d)[:4]:
  k = np.searchsorted(offs, i, side='right')-1
  print(i, names[k], i-offs[k], 'theta0', theta[i], 'gpu', th[i], 'ref', theta_ref[i], 'grad gpu', g[i], 'ref', gref[i])
print('configs equal', np.array_equal(eng.get_configs(), cfg))
print('---- accumulators at the mismatching index (fresh engine, replay)')
session.reset_default_graph(); wavefunctions.reset_name_scope()
wf, ham, opt, ops, sess, shared = T._build('EnergyGradient', hp)
eng = wf._engine
eng.mc_steps(hp.num_equilibration_sweeps * n); eng.reset_accumulators()
for bi in range(hp.num_batches_per_epoch):
  eng.accumulate(0)
  a = eng.get_accumulators()
  cfgb = eng.get_configs()
  logit, zs, acts = vo.fc_logit(theta, cfgb, h, L, dtype=np.float64, return_acts=True)
  print('batch', bi, 'g1[400]', a[400], 'g2[400]', a[p+400], 'oracle z(layer2, unit0) max', zs[1][:,0].max(), 'n alive', (zs[1][:,0]>0).sum())
  eng.mc_steps(n)
