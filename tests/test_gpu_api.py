"""GPU: the reference-API mirror (wavefunctions / operators / graph_builders / training /
evaluation / run scripts) end to end, against epochs restated with the oracle."""
import os

import numpy as np
import pytest

from oracle import vmc_oracle as vo

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _fresh_graph(monkeypatch):
  from cgs_vmc_amd import session, wavefunctions
  session.reset_default_graph()
  wavefunctions.reset_name_scope()
  monkeypatch.setenv('CGS_VMC_SEED', '77')
  monkeypatch.setenv('CGS_VMC_CONFIG_SEED', '5')
  monkeypatch.setenv('CGS_VMC_INIT_SEED', '31')
  yield


def _hparams(**kw):
  from cgs_vmc_amd import utils
  base = dict(wavefunction_type='fully_connected', num_sites=8, num_fc_layers=2, fc_layer_size=16,
              batch_size=32, num_equilibration_sweeps=2, num_monte_carlo_sweeps=1,
              num_batches_per_epoch=3, learning_rates=[1e-2, 1e-3], learning_rate_stops=[1])
  base.update(kw)
  return utils.create_hparams(**base)


def _build(optimizer_name, hp, jx=-1.0):
  from cgs_vmc_amd import operators, session, training, wavefunctions
  from cgs_vmc_amd import lattice
  wf = wavefunctions.build_wavefunction(hp)
  ham = operators.HeisenbergHamiltonian(lattice.chain_bonds(hp.num_sites), jx, 1.)
  opt = training.GROUND_STATE_OPTIMIZERS[optimizer_name]()
  shared = {}
  ops = opt.build_opt_ops(wavefunction=wf, hamiltonian=ham, hparams=hp, shared_resources=shared)
  sess = session.Session()
  sess.run([session.global_variables_initializer(), session.local_variables_initializer()])
  return wf, ham, opt, ops, sess, shared


def _oracle_sweeps(theta, cfg, n_steps, step0, hp):
  return vo.run_sweeps(theta, cfg, n_steps, 77, step0, hp.fc_layer_size, hp.num_fc_layers,
                       dtype=np.float64)[0]


def test_energy_gradient_epoch_matches_oracle_epoch():
  """training.py:589-623 executed by the HIP path == the same epoch restated with the oracle
  (same Philox stream, same initial chains and parameters)."""
  from cgs_vmc_amd import graph_builders
  hp = _hparams()
  wf, ham, opt, ops, sess, shared = _build('EnergyGradient', hp)
  n, h, L, b = hp.num_sites, hp.fc_layer_size, hp.num_fc_layers, hp.batch_size
  theta = wf._get_theta().copy()
  cfg = shared[graph_builders.ResourceName.CONFIGS].eval()
  bonds = ham._bonds_list
  adam = vo.AdamState(theta.size)
  step = 0
  well = np.ones(theta.size, bool)
  for epoch in range(2):
    # --- oracle epoch
    cfg = _oracle_sweeps(theta, cfg, hp.num_equilibration_sweeps * n, step, hp)
    step += hp.num_equilibration_sweeps * n
    acc = vo.Accumulators(theta.size, np.float64)
    for _ in range(hp.num_batches_per_epoch):
      vo.energy_gradient_accumulate(acc, theta, cfg, bonds, -1.0, 1.0, -10.0, h, L, np.float64)
      cfg = _oracle_sweeps(theta, cfg, hp.num_monte_carlo_sweeps * n, step, hp)
      step += hp.num_monte_carlo_sweeps * n
    lr = vo.piecewise_constant(epoch, hp.learning_rate_stops, hp.learning_rates)
    grad = vo.energy_gradient(acc)
    well = well & (np.abs(grad) > 1e-3 * np.abs(grad).max())
    theta = vo.adam_apply(adam, theta, grad, lr, 0.9, hp.beta2, 1e-8)
    # --- HIP epoch
    energy = opt.run_optimization_epoch(ops, sess, hp, epoch)
    assert abs(energy - acc.mean_energy()) < 2e-4 * max(1, abs(acc.mean_energy()))
    np.testing.assert_array_equal(shared[graph_builders.ResourceName.CONFIGS].eval(), cfg)
    # Parameters whose O_k is constant over the batch (b_out always; the bias / w_out entry of a
    # unit that is active on every sample) have an identically-zero covariance gradient: Adam
    # turns their rounding noise into steps of order lr in ANY fp32 implementation.  Only
    # well-conditioned entries (|grad| > 1e-3 max|grad| in every epoch so far) are compared.
    np.testing.assert_allclose(wf._get_theta()[well], theta[well], rtol=0, atol=5e-5)
    assert well.sum() > 0.8 * well.size
  assert sess.run(graph_builders.get_or_create_num_epochs()) == 2


def test_log_overlap_itswo_epoch_matches_oracle_epoch():
  """training.py:731-763 by the HIP path == the oracle restatement (supervisor refreshed once
  per epoch, Adam applied every batch, energy = mean of the LAST batch)."""
  from cgs_vmc_amd import graph_builders
  hp = _hparams(num_batches_per_epoch=2)
  wf, ham, opt, ops, sess, shared = _build('LogOverlapITSWO', hp)
  n, h, L = hp.num_sites, hp.fc_layer_size, hp.num_fc_layers
  theta = wf._get_theta().copy()
  cfg = shared[graph_builders.ResourceName.CONFIGS].eval()
  bonds = ham._bonds_list
  adam = vo.AdamState(theta.size)
  step = 0
  well = np.ones(theta.size, bool)
  for epoch in range(2):
    cfg = _oracle_sweeps(theta, cfg, hp.num_equilibration_sweeps * n, step, hp)
    step += hp.num_equilibration_sweeps * n
    theta_w = theta.copy()                                   # update_supervisor
    lr = vo.piecewise_constant(epoch, hp.learning_rate_stops, hp.learning_rates)
    for _ in range(hp.num_batches_per_epoch):
      cfg = _oracle_sweeps(theta, cfg, hp.num_monte_carlo_sweeps * n, step, hp)
      step += hp.num_monte_carlo_sweeps * n
      acc = vo.Accumulators(theta.size, np.float64)          # reset_gradients
      vo.log_overlap_accumulate(acc, theta, theta_w, cfg, bonds, -1.0, 1.0, -10.0, -10.0,
                                hp.time_evolution_beta, h, L, np.float64)
      grad = vo.log_overlap_gradient(acc)
      well = well & (np.abs(grad) > 1e-3 * np.abs(grad).max())
      theta = vo.adam_apply(adam, theta, grad, lr, 0.9, hp.beta2, 1e-8)
    energy = opt.run_optimization_epoch(ops, sess, hp, epoch)
    assert abs(energy - acc.mean_energy()) < 2e-4 * max(1, abs(acc.mean_energy()))
    np.testing.assert_array_equal(shared[graph_builders.ResourceName.CONFIGS].eval(), cfg)
    np.testing.assert_allclose(wf._get_theta()[well], theta[well], rtol=0, atol=1e-4)
    assert well.sum() > 0.8 * well.size


def test_tensor_handles_match_oracle():
  """wavefunction(configs), Operator.build / local_value / apply_in_place as op handles."""
  from cgs_vmc_amd import graph_builders
  hp = _hparams()
  wf, ham, opt, ops, sess, shared = _build('EnergyGradient', hp)
  configs = shared[graph_builders.ResourceName.CONFIGS]
  cfg = configs.eval()
  theta = wf._get_theta()
  h, L = hp.fc_layer_size, hp.num_fc_layers
  amp = lambda c: vo.fc_psi(theta, c, h, L, dtype=np.float64)
  psi = sess.run(wf(configs))
  np.testing.assert_allclose(psi, amp(cfg), rtol=3e-5)
  np.testing.assert_allclose(sess.run(wf(cfg[:5])), amp(cfg[:5]), rtol=3e-5)
  diag, off = sess.run(list(ham.build(wf, configs)))
  dref, oref = vo.heisenberg_build(amp, cfg, ham._bonds_list, -1.0, 1.0, np.float64)
  np.testing.assert_allclose(diag, dref, atol=1e-6)
  np.testing.assert_allclose(off, oref, rtol=3e-4, atol=1e-3)
  lv = sess.run(ham.local_value(wf, configs))
  np.testing.assert_allclose(lv, dref + oref / amp(cfg), rtol=2e-4, atol=2e-4)
  ap = sess.run(ham.apply_in_place(wf, configs))
  np.testing.assert_allclose(ap, dref * amp(cfg) + oref, rtol=3e-4, atol=1e-3)
  with pytest.raises(ValueError):
    wf(np.ones((3, hp.num_sites + 1), np.float32))
  # one mc_step through Session.run moves the chains and reports the acceptance count
  sess.run(ops.mc_step)
  acc = sess.run(ops.acc_rate)
  assert 0 <= acc <= hp.batch_size
  assert ((configs.eval() != cfg).any(1).sum()) == acc


def test_evaluator_returns_batch_means():
  from cgs_vmc_amd import evaluation, lattice, operators, session, wavefunctions
  hp = _hparams(num_evaluation_samples=4)
  wf = wavefunctions.build_wavefunction(hp)
  ham = operators.HeisenbergHamiltonian(lattice.chain_bonds(hp.num_sites), -1.0, 1.)
  ev = evaluation.MonteCarloOperatorEvaluator()
  shared = {}
  eops = ev.build_eval_ops(wavefunction=wf, operator=ham, hparams=hp, shared_resources=shared)
  sess = session.Session()
  sess.run(session.global_variables_initializer())
  theta = wf._get_theta()
  from cgs_vmc_amd import graph_builders
  cfg = shared[graph_builders.ResourceName.CONFIGS].eval()
  n, h, L = hp.num_sites, hp.fc_layer_size, hp.num_fc_layers
  amp = lambda c: vo.fc_psi(theta, c, h, L, dtype=np.float64)
  cfg = _oracle_sweeps(theta, cfg, hp.num_equilibration_sweeps * n, 0, hp)
  step = hp.num_equilibration_sweeps * n
  ref = []
  for _ in range(4):
    ref.append(vo.local_value(amp, cfg, ham._bonds_list, -1.0, 1.0, dtype=np.float64).mean())
    cfg = _oracle_sweeps(theta, cfg, n, step, hp)
    step += n
  vals = ev.run_evaluation(eops, sess, hp, epoch_num=0)
  assert len(vals) == 4
  np.testing.assert_allclose(vals, ref, rtol=2e-4, atol=2e-4)


def test_run_training_and_energy_evaluation_cli(tmp_path, capsys):
  """BASELINE config 1 (plumbing): 16-site chain and 4x4 torus via J.txt, B=64, 2x32 ansatz."""
  from cgs_vmc_amd import lattice, run_energy_evaluation, run_training, session, wavefunctions
  for name, bonds in (('chain', None), ('torus', lattice.torus_bonds(4, 4))):
    session.reset_default_graph(); wavefunctions.reset_name_scope()
    d = str(tmp_path / name)
    os.makedirs(d)
    if bonds:
      lattice.write_bonds(d, bonds)
    hp = ('batch_size=64,fc_layer_size=32,num_fc_layers=2,num_equilibration_sweeps=5,'
          'num_batches_per_epoch=10,learning_rates=[0.003,0.001],learning_rate_stops=[100],'
          'num_evaluation_samples=10')
    run_training.main(['--checkpoint_dir', d, '--num_sites', '16', '--heisenberg_jx', '-1.0',
                       '--wavefunction_type', 'fully_connected', '--optimizer', 'EnergyGradient',
                       '--num_epochs', '12', '--hparams', hp])
    metrics = [float(x) for x in open(os.path.join(d, 'metrics.txt')).read().split()]
    assert len(metrics) == 12 and np.isfinite(metrics).all()
    assert metrics[-1] < metrics[0] - 0.3            # the energy goes down
    assert os.path.exists(os.path.join(d, 'hparams.pbtxt'))
    assert os.path.exists(os.path.join(d, 'model_prior_11_epochs.npz'))
    assert not os.path.exists(os.path.join(d, 'model_prior_0_epochs.npz'))     # max_to_keep=5
    session.reset_default_graph(); wavefunctions.reset_name_scope()
    mean, unc = run_energy_evaluation.main(['--checkpoint_dir', d, '--heisenberg_jx', '-1.0'])
    out = capsys.readouterr().out
    assert 'Energy: ' in out and ' +/- ' in out
    # evaluation of the last checkpoint (parameters BEFORE epoch 11) ~ the energy of epoch 10/11
    assert abs(mean - metrics[-1]) < 0.6


def test_accumulator_device_view_for_rccl():
  """parallel.accumulator_tensor: zero-copy torch view of the library's device buffer."""
  import torch
  from cgs_vmc_amd import _hip, parallel
  from cgs_vmc_amd.engine import VmcEngine
  n, h, L, b = 8, 16, 2, 32
  eng = VmcEngine(n, b, L, h)
  eng.set_params(vo.init_params(n, h, L, np.random.default_rng(0)))
  eng.set_configs(vo.random_configurations(n, b, np.random.RandomState(0)))
  eng.set_bonds(vo.chain_bonds(n), -1.0, 1.0)
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  eng.synchronize()
  t = parallel.accumulator_tensor(eng)
  assert t.is_cuda and t.dtype == torch.float32 and t.numel() == 2 * eng.num_params + 8
  np.testing.assert_array_equal(t.cpu().numpy(), eng.get_accumulators())
  t.mul_(2.0)                                   # in place on the library's memory
  torch.cuda.synchronize()
  np.testing.assert_array_equal(eng.get_accumulators(), t.cpu().numpy())
  parallel.allreduce_accumulators(eng)          # world size 1: no-op
  eng.close()


# (ranks, --collective, CGS_VMC_TRANSPORT).  4 ranks is the widest multi-process job a box of this pool
# can hold (at most 6 processes on the GPU, pytest itself is one); the default N > 1 step is the
# library's one-call entry, its default transport under gloo the host hook, under nccl the device hook.
@pytest.mark.parametrize('gpus,collective,transport', [(2, None, None), (4, 'library', 'torch'), (2, 'torch', None)])
def test_multi_rank_bench_path_on_one_gpu(tmp_path, gpus, collective, transport):
  """`python bench.py --gpus N` with NO rank environment: bench.py starts the N ranks itself
  (sharded chains, accumulator all-reduce, max-over-ranks timing); they share this GPU over gloo.
  RCCL itself needs >= 2 GPUs and is exercised by the driver's scaling runs; everything around
  the collective is covered here."""
  import json
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = {k: v for k, v in os.environ.items()
         if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'CGS_VMC_TRANSPORT')}
  env['CGS_VMC_DIST_BACKEND'] = 'gloo'
  if transport:
    env['CGS_VMC_TRANSPORT'] = transport
  p = subprocess.run(
      [sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(gpus), '--steps', '2', '--warmup',
       '1', '--reps', '2', '--no-cpu-baseline', '--workload', 'heisenberg6x6_fc3x128_b1024']
      + (['--collective', collective] if collective else []),
      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
  assert p.returncode == 0, p.stderr.decode()[-2000:]
  lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
  assert len(lines) == 1                      # rank 0 only
  d = json.loads(lines[0])
  assert d['n_gpus'] == gpus and d['config']['global_chains'] == gpus * 1024 and d['value'] > 0
  r = d['rccl']
  assert r['ranks'] == gpus and r['allreduce_floats'] == 2 * 37889 + 8
  # the N > 1 line proves what it ran on and that its collectives reduce
  assert r['backend'] == 'gloo'
  assert r['library_transport'].startswith('device hook' if transport == 'torch' else 'host hook')
  assert [i['rank'] for i in r['devices']] == list(range(gpus))
  assert all(i['ordinal'] == 0 and i['pci_bus_id'] and i['name'] for i in r['devices'])
  assert r['distinct_devices'] is False        # the ranks share this GPU: tolerated under gloo only
  # the device identity comes from the runtime the process already has -- every rank
  # maps exactly ONE libamdhip64, and it is the one behind the library's hip* symbols
  for i in r['devices']:
    assert len(i['hip_runtimes_mapped']) == 1, i['hip_runtimes_mapped']
    assert os.path.realpath(i['hip_runtime']) == i['hip_runtimes_mapped'][0]
  c = r['checked_allreduce']
  assert c['ok'] is True and c['library_ok'] is True and c['process_group_on_accumulators']
  assert c['library_sum_of_ranks'] == c['expected_sum_of_ranks'] == gpus * (gpus - 1) / 2
  assert c['library_sum_of_ones'] == gpus and c['library_max_of_ranks'] == gpus - 1
  assert r['allreduce_ms_blocking'] > 0 and r['collective_fallback'] is None
  # the timed step goes through the library's own entry by default, and both ways
  # of taking the step leave the same accumulators
  assert r['timed_collective'].startswith('torch.distributed' if collective == 'torch' else 'library entry')
  assert ('in stream' in d['config']['step']) == (collective != 'torch')
  pa = r['paths_agree']
  assert pa['g_count'] == [1.0, 1.0] and pa['e_count'] == [gpus * 1024.0] * 2
  assert pa['bitwise'] if gpus == 2 else pa['max_abs_diff_over_max_abs'] < 1e-6
  m = r['ms_per_step_ranks']
  assert len(m['all']) == gpus and 0 < m['min'] <= m['max'] <= d['ms_per_step'] * 1.05
  if gpus != 2 or collective:
    return
  # a rank count that contradicts the environment is refused instead of silently mislabelled
  bad = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1'],
                       env=dict(env, WORLD_SIZE='1', RANK='0'), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
  assert bad.returncode == 2 and b'WORLD_SIZE' in bad.stderr


def test_one_rank_through_the_launcher_costs_nothing():
  """`bench.py --gpus 1 --spawn` (the rank started by bench.py's own launcher, as for N > 1) against
  the direct N = 1 run: same value within 1 % (medians of 5 repetitions of 50 steps at config 3;
  one re-measurement allowed, run-to-run spread of a median is ~0.5 %)."""
  import json
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = {k: v for k, v in os.environ.items()
         if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}

  def run(extra):
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '50', '--warmup', '5',
                        '--no-cpu-baseline', '--no-timing', '--no-extra'] + extra, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return json.loads([l for l in p.stdout.decode().splitlines() if l.startswith('{')][-1])

  rel = None
  for _ in range(2):
    direct, spawned = run([]), run(['--gpus', '1', '--spawn'])
    assert direct['n_gpus'] == spawned['n_gpus'] == 1 and 'rccl' not in spawned
    rel = abs(spawned['value'] - direct['value']) / direct['value']
    if rel <= 0.01:
      break
  assert rel <= 0.01, (direct['ms_per_step'], spawned['ms_per_step'])


@pytest.mark.parametrize('optimizer', ['EnergyGradient', 'LogOverlapITSWO'])
def test_training_reaches_exact_ground_state_energy(tmp_path, optimizer):
  """End-to-end physics check: 400 epochs of either working optimizer on the 4x4 Heisenberg
  torus (Marshall-rotated, j_x = -1) bring the VMC energy within 1 % of the exact-
  diagonalisation ground state energy E0 = -11.2285 (16 sites, E0/N = -0.70178)."""
  from cgs_vmc_amd import lattice, run_training
  d = str(tmp_path)
  lattice.write_bonds(d, lattice.torus_bonds(4, 4))
  hp = ('batch_size=512,fc_layer_size=64,num_fc_layers=2,num_equilibration_sweeps=10,'
        'num_batches_per_epoch=20,learning_rates=[0.003,0.001,0.0003],'
        'learning_rate_stops=[150,300]')
  run_training.main(['--checkpoint_dir', d, '--num_sites', '16', '--heisenberg_jx', '-1.0',
                     '--wavefunction_type', 'fully_connected', '--optimizer', optimizer,
                     '--num_epochs', '400', '--hparams', hp])
  energies = [float(x) for x in open(os.path.join(d, 'metrics.txt')).read().split()]
  tail = np.mean(energies[-20:])
  assert abs(tail - (-11.2285)) < 0.01 * 11.2285, tail
  assert tail > -11.2285 - 0.05          # variational: not below E0 beyond MC noise


def test_resume_training_restores_latest_checkpoint(tmp_path):
  """--resume_training (run_training.py:141-143): the second run starts from the parameters of
  the latest checkpoint (the ones PRIOR to the first run's last epoch) and keeps appending to
  metrics.txt."""
  from cgs_vmc_amd import run_training, session, wavefunctions
  d = str(tmp_path)
  hp = ('batch_size=64,fc_layer_size=32,num_fc_layers=2,num_equilibration_sweeps=3,'
        'num_batches_per_epoch=5')
  common = ['--checkpoint_dir', d, '--num_sites', '16', '--heisenberg_jx', '-1.0',
            '--wavefunction_type', 'fully_connected', '--optimizer', 'EnergyGradient',
            '--hparams', hp]
  run_training.main(common + ['--num_epochs', '4'])
  first = np.load(os.path.join(d, 'model_prior_3_epochs.npz'))
  session.reset_default_graph(); wavefunctions.reset_name_scope()
  run_training.main(common + ['--num_epochs', '2', '--resume_training'])
  # the resumed run wrote its own 'prior to epoch 0' checkpoint = the restored parameters
  resumed = np.load(os.path.join(d, 'model_prior_0_epochs.npz'))
  assert sorted(first.files) == sorted(resumed.files) and len(first.files) == 6
  for k in first.files:
    np.testing.assert_array_equal(first[k], resumed[k])
  metrics = open(os.path.join(d, 'metrics.txt')).read().split()
  assert len(metrics) == 6


def test_update_norm_keeps_psi_finite_during_training():
  """Wavefunction.update_norm through the op handle: after the shift update max psi <= 1e10
  and ratios (sampling, local energy) are unchanged."""
  from cgs_vmc_amd import graph_builders
  hp = _hparams()
  wf, ham, opt, ops, sess, shared = _build('EnergyGradient', hp)
  configs = shared[graph_builders.ResourceName.CONFIGS]
  e_before = sess.run(ham.local_value(wf, configs))
  wf._set_shift(-45.0)                              # psi = exp(logit + 45) ~ 3e19 > 1e10
  assert sess.run(wf(configs)).max() > 1e10
  sess.run(ops.update_wf_norm)
  psi = sess.run(wf(configs))
  assert np.isfinite(psi).all() and abs(psi.max() - 1e10) < 1e-3 * 1e10
  np.testing.assert_allclose(sess.run(ham.local_value(wf, configs)), e_before, rtol=1e-6)


def test_rccl_single_rank_allreduce_on_library_buffer():
  """RCCL itself on this box (1 rank): see tests/_rccl_worker.py."""
  import socket
  import subprocess
  import sys
  s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1',
             LOCAL_RANK='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
  p = subprocess.run([sys.executable, os.path.join(root, 'tests', '_rccl_worker.py')], env=env,
                     stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
  assert p.returncode == 0 and b'rccl ok' in p.stdout, p.stdout.decode()[-3000:]


def test_cabi_allreduce_over_an_rccl_communicator():
  """vmc_allreduce_accumulators with a communicator made directly on librccl (1 rank): the
  collective runs on the engine's stream; SUM over one rank is the identity and g_count is
  divided by the world size the caller states."""
  import ctypes as C
  from cgs_vmc_amd.engine import VmcEngine
  # the communicator has to come from the librccl instance the library calls into (with torch in the
  # process that is torch's bundled copy, not necessarily the first librccl.so on the loader path)
  from cgs_vmc_amd import _hip as hip_binding
  path = hip_binding.load().vmc_rccl_library_path().decode()
  assert path and os.path.exists(path), path
  rccl = C.CDLL(path)

  class UniqueId(C.Structure):
    _fields_ = [('internal', C.c_char * 128)]

  uid = UniqueId()
  assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
  comm = C.c_void_p()
  rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
  assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
  try:
    n, h, L, b = 16, 32, 2, 64
    rng = np.random.default_rng(0)
    eng = VmcEngine(n, b, L, h, seed=5)
    eng.set_params(vo.init_params(n, h, L, rng))
    eng.set_configs(vo.random_configurations(n, b, np.random.RandomState(1)))
    eng.set_bonds(vo.torus_bonds(4, 4), -1.0, 1.0)
    eng.reset_accumulators()
    eng.accumulate(0)
    before = eng.get_accumulators()
    eng.allreduce_accumulators_rccl(0, 1)                       # NULL communicator, one rank: no-op
    np.testing.assert_array_equal(eng.get_accumulators(), before)
    from cgs_vmc_amd import _hip
    with pytest.raises(_hip.HipLibraryError):                   # ranks > 1 without any transport
      eng.allreduce_accumulators_rccl(0, 8)
    np.testing.assert_array_equal(eng.get_accumulators(), before)
    eng.allreduce_accumulators_rccl(comm.value, 2)             # really calls ncclAllReduce
    after = eng.get_accumulators()
    expect = before.copy()
    expect[-4] /= 2                                            # g_count / stated world size
    np.testing.assert_array_equal(after, expect)
    eng.close()
  finally:
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)


def test_heisenberg_bond_operator_and_interleaved_operators():
  """operators.HeisenbergBond (operators.py:128-169) as the one-bond case, evaluated in turn
  with the full Hamiltonian on the same CONFIGS variable (each handle re-applies its own bond
  set when it is run)."""
  from cgs_vmc_amd import graph_builders, operators
  hp = _hparams()
  wf, ham, opt, ops, sess, shared = _build('EnergyGradient', hp)
  configs = shared[graph_builders.ResourceName.CONFIGS]
  cfg = configs.eval()
  theta = wf._get_theta()
  h, L = hp.fc_layer_size, hp.num_fc_layers
  amp = lambda c: vo.fc_psi(theta, c, h, L, dtype=np.float64)
  bond = operators.HeisenbergBond((2, 7), 0.8, 1.3)
  t_bond = bond.build(wf, configs)
  t_ham = ham.build(wf, configs)
  lv_ham = ham.local_value(wf, configs)
  for _ in range(2):                                   # interleaved evaluation
    d1, o1 = sess.run(list(t_bond))
    dref, oref = vo.heisenberg_build(amp, cfg, [(2, 7)], 0.8, 1.3, np.float64)
    np.testing.assert_allclose(d1, dref, atol=1e-6)
    np.testing.assert_allclose(o1, oref, rtol=3e-4, atol=1e-3)
    d2, o2 = sess.run(list(t_ham))
    dref2, oref2 = vo.heisenberg_build(amp, cfg, ham._bonds_list, -1.0, 1.0, np.float64)
    np.testing.assert_allclose(d2, dref2, atol=1e-6)
    np.testing.assert_allclose(o2, oref2, rtol=3e-4, atol=1e-3)
    np.testing.assert_allclose(sess.run(lv_ham), dref2 + oref2 / amp(cfg), rtol=2e-4, atol=2e-4)
  np.testing.assert_allclose(sess.run(bond.local_value(wf, configs)), dref + oref / amp(cfg),
                             rtol=2e-4, atol=2e-4)
