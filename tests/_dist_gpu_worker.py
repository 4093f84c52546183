"""Worker of tests/test_gpu_dist.py: WORLD_SIZE (2 or 4) gloo ranks sharing ONE GPU drive the product
routing for sharded chains (training.run_optimization_epoch -> vmc_epoch_*_dist, parallel.sr_solve ->
vmc_sr_solve_dist, evaluation.run_evaluation -> vmc_evaluate) and compare every epoch with an
UNSHARDED engine stepping the same global batch op by op.  Transport (CGS_VMC_TRANSPORT): 'host' =
the host all-reduce hook, or 'torch' = the device hook, i.e. torch.distributed.all_reduce on the
library's device buffer in stream order -- the default transport of a real multi-GPU job (backend
'nccl'), here over gloo, which reduces device tensors too (RCCL refuses two ranks on one device)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from cgs_vmc_amd import _hip, graph_builders, lattice, operators, parallel, session, training, utils, wavefunctions  # noqa: E402
from cgs_vmc_amd.engine import VmcEngine  # noqa: E402


def _gather_rows(local_rows):
  t = torch.from_numpy(np.ascontiguousarray(local_rows, np.float32))
  parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
  dist.all_gather(parts, t)
  return np.concatenate([p.numpy() for p in parts])


def main():
  os.environ.update(CGS_VMC_SEED='77', CGS_VMC_CONFIG_SEED='5', CGS_VMC_INIT_SEED='31')
  parallel.init_from_env('gloo')
  world = int(os.environ['WORLD_SIZE'])
  assert parallel.world_size() == world
  rank = parallel.rank()
  coll = parallel.collective()
  want = os.environ.get('CGS_VMC_TRANSPORT', 'host')
  assert coll.comm == 0 and coll.transport == want
  assert (coll.device_hook() is not None) == (want == 'torch') and (coll.host_hook() is not None) == (want == 'host')
  lb = 64 // world
  for name in ('LogOverlapITSWO', 'EnergyGradient', 'StochasticReconfiguration'):
    session.reset_default_graph()
    wavefunctions.reset_name_scope()
    hp = utils.create_hparams(wavefunction_type='fully_connected', num_sites=16, num_fc_layers=2,
                              fc_layer_size=32, batch_size=64, num_equilibration_sweeps=2,
                              num_monte_carlo_sweeps=1, num_batches_per_epoch=3,
                              learning_rates=[1e-2, 1e-3], learning_rate_stops=[1])
    n, h, L, nb = hp.num_sites, hp.fc_layer_size, hp.num_fc_layers, hp.num_batches_per_epoch
    wf = wavefunctions.build_wavefunction(hp)
    ham = operators.HeisenbergHamiltonian(lattice.chain_bonds(n), -1.0, 1.0)
    opt = training.GROUND_STATE_OPTIMIZERS[name]()
    shared = {}
    ops = opt.build_opt_ops(wavefunction=wf, hamiltonian=ham, hparams=hp, shared_resources=shared)
    sess = session.Session()
    sess.run([session.global_variables_initializer(), session.local_variables_initializer()])
    cfg_var = shared[graph_builders.ResourceName.CONFIGS]
    assert cfg_var.local_batch == lb and cfg_var.chain_offset == lb * rank
    # the unsharded twin: the gathered global batch on one engine, stepped op by op
    ref = VmcEngine(n, 64, L, h, seed=77)
    ref.set_params(wf._get_theta())
    ref.set_configs(_gather_rows(cfg_var.eval()))
    ref.set_bonds(ham._bonds_list, -1.0, 1.0)
    if name == 'StochasticReconfiguration':
      ref.sr_reserve(nb)
    well = np.ones(ref.num_params, bool)
    for epoch in range(2):
      lr = training.piecewise_constant(epoch, [1], [1e-2, 1e-3])
      ref.mc_steps(2 * n)
      ref.update_norm(1e10)
      if name == 'LogOverlapITSWO':
        ref.transfer_params()
        for _ in range(nb):
          ref.mc_steps(n)
          ref.reset_accumulators()
          ref.accumulate(_hip.VMC_MODE_LOG_OVERLAP_ITSWO, hp.time_evolution_beta)
          g = ref.get_gradient(_hip.VMC_MODE_LOG_OVERLAP_ITSWO)
          well &= np.abs(g) > 1e-3 * np.abs(g).max()
          ref.apply_adam(_hip.VMC_MODE_LOG_OVERLAP_ITSWO, lr, 0.9, hp.beta2, 1e-8)
        e_ref = ref.mean_energy()
      else:
        ref.reset_accumulators()
        for _ in range(nb):
          ref.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
          ref.mc_steps(n)
        e_ref = ref.mean_energy()
        if name == 'EnergyGradient':
          g = ref.get_gradient(_hip.VMC_MODE_ENERGY_GRADIENT)
          well &= np.abs(g) > 1e-3 * np.abs(g).max()
          ref.apply_adam(_hip.VMC_MODE_ENERGY_GRADIENT, lr, 0.9, hp.beta2, 1e-8)
        else:
          it_ref, res_ref = ref.sr_solve(0.01, 1e-3, 100)
          ref.sr_apply(lr)
        ref.reset_accumulators()
      energy = opt.run_optimization_epoch(ops, sess, hp, epoch)
      assert abs(energy - e_ref) < 2e-5 * max(1.0, abs(e_ref)), (name, epoch, energy, e_ref)
      np.testing.assert_array_equal(cfg_var.eval(), ref.get_configs()[lb * rank:lb * (rank + 1)])
      got, want = wf._get_theta(), ref.get_params()
      if name == 'StochasticReconfiguration':
        assert opt.last_cg[0] == it_ref and abs(opt.last_cg[1] - res_ref) < 1e-3, (opt.last_cg, it_ref, res_ref)
        assert np.abs(got - want).max() < 5e-5, np.abs(got - want).max()
      else:
        assert np.abs(got - want)[well].max() < 5e-5 and well.sum() > 0.7 * well.size, \
            (name, epoch, np.abs(got - want)[well].max(), well.sum())
      assert abs(wf._get_shift() - ref.get_shift()) == 0.0
      # keep both trajectories on the same parameters / Adam moments epoch by epoch
      ref.set_params(got)
      m, v, t = cfg_var._engine.get_adam_state()
      ref.set_adam_state(m, v, t)
    every = _gather_rows(wf._get_theta()[None, :])
    for r in range(1, world):
      np.testing.assert_array_equal(every[0], every[r])   # identical step on every rank
    if name == 'EnergyGradient':
      # evaluation.run_evaluation on the sharded chains: the fused entry (vmc_evaluate, one float64
      # all-reduce) against the op-by-op loop (one all-reduce per sample) and the unsharded engine
      from cgs_vmc_amd import evaluation
      hp.set_hparam('num_evaluation_samples', 5)
      ev = evaluation.MonteCarloOperatorEvaluator()
      eops = ev.build_eval_ops(wavefunction=wf, operator=ham, hparams=hp, shared_resources=shared)
      eng = cfg_var._engine
      start, step0 = cfg_var.eval().copy(), eng.step_counter
      fused = ev.run_evaluation(eops, sess, hp, 0)
      acc_fused = ev.acceptance_count
      cfg_var.load(start); eng.step_counter = step0
      os.environ['CGS_VMC_EVAL_FUSED'] = '0'
      loop = ev.run_evaluation(eops, sess, hp, 0)
      del os.environ['CGS_VMC_EVAL_FUSED']
      assert len(fused) == 5 and fused == loop and acc_fused == ev.acceptance_count, (fused, loop)
      ref.set_params(wf._get_theta())
      ref.set_shift(wf._get_shift())
      ref.set_configs(_gather_rows(start)); ref.step_counter = step0
      m_ref, _ = ref.evaluate(None, 2 * n, 5, n)
      assert np.allclose(fused, m_ref, rtol=1e-6, atol=1e-6), (fused, m_ref)
    ref.close()
  dist.barrier()
  dist.destroy_process_group()
  print('rank {} ok'.format(rank))


if __name__ == '__main__':
  main()
