"""MonteCarloOperatorEvaluator.run_evaluation (evaluation.py:113-152) wall time: the op-by-op loop (two host
calls per sample: CGS_VMC_EVAL_FUSED=0) against the one-call device-resident entry vmc_evaluate, at BASELINE
configs 1 and 2 with the reference's default evaluation hparams (100 equilibration sweeps, 100 samples one
sweep apart).  The two lists of means are compared bit for bit."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(CGS_VMC_SEED='77', CGS_VMC_CONFIG_SEED='5', CGS_VMC_INIT_SEED='31')
from cgs_vmc_amd import evaluation, graph_builders, lattice, operators, session, utils, wavefunctions  # noqa: E402

CONFIGS = {
    'config1_chain16_fc2x32_b64': dict(num_sites=16, num_fc_layers=2, fc_layer_size=32, batch_size=64, bonds='chain'),
    'config2_6x6_fc3x128_b1024': dict(num_sites=36, num_fc_layers=3, fc_layer_size=128, batch_size=1024, bonds='torus6'),
}
out = {}
for name, cfg in CONFIGS.items():
  bonds = lattice.chain_bonds(cfg['num_sites']) if cfg['bonds'] == 'chain' else lattice.torus_bonds(6, 6, False)
  res = {}
  for mode in ('loop', 'fused'):
    session.reset_default_graph()
    wavefunctions.reset_name_scope()
    hp = utils.create_hparams(wavefunction_type='fully_connected', num_sites=cfg['num_sites'],
                              num_fc_layers=cfg['num_fc_layers'], fc_layer_size=cfg['fc_layer_size'],
                              batch_size=cfg['batch_size'])
    wf = wavefunctions.build_wavefunction(hp)
    ham = operators.HeisenbergHamiltonian(bonds, -1.0, 1.0)
    shared = {}
    ev = evaluation.MonteCarloOperatorEvaluator()
    ops = ev.build_eval_ops(wavefunction=wf, operator=ham, hparams=hp, shared_resources=shared)
    sess = session.Session()
    sess.run([session.global_variables_initializer(), session.local_variables_initializer()])
    cfg_var = shared[graph_builders.ResourceName.CONFIGS]
    eng = cfg_var._engine
    start, step0 = cfg_var.eval().copy(), eng.step_counter
    os.environ['CGS_VMC_EVAL_FUSED'] = '1' if mode == 'fused' else '0'
    ev.run_evaluation(ops, sess, hp, 0)                       # warm-up (caches, clocks)
    times = []
    for _ in range(5):
      cfg_var.load(start); eng.step_counter = step0
      eng.synchronize()
      t0 = time.perf_counter()
      vals = ev.run_evaluation(ops, sess, hp, 0)
      times.append(time.perf_counter() - t0)
    res[mode] = {'ms': 1e3 * sorted(times)[2], 'values': [float(v) for v in vals], 'accepted': ev.acceptance_count}
  same = res['loop']['values'] == res['fused']['values'] and res['loop']['accepted'] == res['fused']['accepted']
  out[name] = {'samples': hp.num_evaluation_samples, 'equilibration_sweeps': hp.num_equilibration_sweeps,
               'loop_ms': res['loop']['ms'], 'fused_ms': res['fused']['ms'], 'speedup': res['loop']['ms'] / res['fused']['ms'],
               'means_bit_identical': bool(same), 'mean_energy_per_site': sum(res['fused']['values']) / len(res['fused']['values']) / cfg['num_sites']}
print(json.dumps(out, indent=1))
