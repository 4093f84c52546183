"""CPU tests of the host-side mirror of the reference API (no GPU compute)."""
import os

import numpy as np
import pytest

from cgs_vmc_amd import (evaluation, graph_builders, lattice, layers, operators, session,
                         training, utils, wavefunctions)
from oracle import vmc_oracle as vo


@pytest.fixture(autouse=True)
def _fresh_graph():
  session.reset_default_graph()
  wavefunctions.reset_name_scope()
  yield


def test_hparams_defaults_match_reference():
  hp = utils.create_hparams()
  # utils.py:87-148
  assert hp.num_sites == 40 and hp.num_fc_layers == 3 and hp.fc_layer_size == 80
  assert hp.nonlinearity == 'relu' and hp.output_activation == 'exp'
  assert hp.num_equilibration_sweeps == 100 and hp.num_monte_carlo_sweeps == 1
  assert hp.batch_size == 200 and hp.num_batches_per_epoch == 50 and hp.num_epochs == 500
  assert hp.time_evolution_beta == 0.12 and hp.beta2 == 0.99 and hp.optimizer == 'adam'
  assert hp.learning_rates == [1e-3, 1e-4, 2e-5, 1e-5]
  assert hp.learning_rate_stops == [300, 600, 1000]
  assert hp.num_evaluation_samples == 100


def test_hparams_parse_set_and_errors():
  hp = utils.create_hparams(batch_size=64)
  hp.parse('num_sites=16,fc_layer_size=32,learning_rates=[0.01,0.001],nonlinearity=relu')
  assert hp.num_sites == 16 and hp.fc_layer_size == 32 and hp.learning_rates == [0.01, 0.001]
  hp.set_hparam('wavefunction_type', 'fully_connected')
  with pytest.raises(ValueError):
    hp.parse('not_a_param=3')
  with pytest.raises(KeyError):
    hp.set_hparam('nope', 1)
  with pytest.raises(ValueError):
    hp.set_hparam('learning_rates', 0.1)      # list-valued


def test_hparams_pbtxt_roundtrip(tmp_path):
  hp = utils.create_hparams()
  hp.parse('batch_size=64,checkpoint_dir=/tmp/x,learning_rates=[0.01,1e-05]')
  path = tmp_path / 'hparams.pbtxt'
  path.write_text(str(hp.to_proto()))
  text = path.read_text()
  assert 'hparam {\n  key: "batch_size"\n  value {\n    int64_value: 64\n  }\n}' in text
  assert 'float_value: 0.99' in text and 'value: 1e-05' in text
  hp2 = utils.load_hparams(str(path))
  for k, v in hp.values().items():
    got = getattr(hp2, k)
    if isinstance(v, float) or (isinstance(v, list) and isinstance(v[0], float)):
      np.testing.assert_allclose(got, v, rtol=1e-6)
    else:
      assert got == v, k


def test_random_configurations_sz0():
  cfg = utils.random_configurations(10, 50)
  assert cfg.dtype == np.float32 and cfg.shape == (50, 10)
  assert (np.abs(cfg) == 1).all() and (cfg.sum(1) == 0).all()
  odd = utils.random_configurations(7, 5)
  assert ((odd == -1).sum(1) == 3).all()
  np.testing.assert_array_equal(utils.random_configurations(8, 4, seed=3),
                                vo.random_configurations(8, 4, np.random.RandomState(3)))


def test_piecewise_constant_and_optimizer_factory():
  f = lambda x: training.piecewise_constant(x, [300, 600, 1000], [1e-3, 1e-4, 2e-5, 1e-5])
  assert [f(0), f(300), f(301), f(600), f(1000), f(1001)] == [1e-3, 1e-3, 1e-4, 1e-4, 2e-5, 1e-5]
  hp = utils.create_hparams()
  opt = training.create_sgd_optimizer(hp)
  assert (opt.beta1, opt.beta2, opt.epsilon) == (0.9, 0.99, 1e-8)
  assert opt.learning_rate() == 1e-3
  graph_builders.get_or_create_num_epochs().value = 301
  assert opt.learning_rate() == 1e-4
  hp.set_hparam('optimizer', 'momentum')
  with pytest.raises(TypeError):        # defect B2 of the reference: same outcome
    training.create_sgd_optimizer(hp)


def test_registries_and_errors():
  # the reference's three (training.py:913-917) + the SR extension named by the north star
  assert set(training.GROUND_STATE_OPTIMIZERS) == {
      'EnergyGradient', 'LogOverlapITSWO', 'ITSWO', 'StochasticReconfiguration'}
  assert set(wavefunctions.WAVEFUNCTION_TYPES) == {
      'fully_connected', 'rbm', 'conv_1d', 'conv_2d', 'mps', 'pbdg', 'fully_connected_nnb',
      'res_net_1d', 'res_net_2d', 'ed_vector', 'gnn'}
  assert set(layers.NONLINEARITIES) == {'relu', 'exp', 'cos', 'tan', 'tanh', 'sigmoid', 'identity'}
  hp = utils.create_hparams(wavefunction_type='bogus')
  with pytest.raises(ValueError, match='not registered'):
    wavefunctions.build_wavefunction(hp)
  hp.set_hparam('wavefunction_type', 'mps')
  with pytest.raises(NotImplementedError):
    wavefunctions.build_wavefunction(hp)
  hp.set_hparam('wavefunction_type', 'conv_1d')
  wf1 = wavefunctions.build_wavefunction(hp)
  wf1._n_sites = 12
  assert wf1._shapes() == (
      ['conv_1d_network/conv_1d_periodic%s/conv_1d/%s' % (sfx, v) for sfx in ('', '_1', '_2', '_3', '_4')
       for v in ('w', 'b')],
      [(5, 1, 16), (16,)] + [(5, 16, 16), (16,)] * 4)
  assert wf1._engine_spec()['ansatz'] == 'conv_1d' and wf1._engine_spec()['kernel_size'] == 5
  hp.set_hparam('wavefunction_type', 'res_net_1d')
  wf1 = wavefunctions.build_wavefunction(hp)
  wf1._n_sites = 12
  assert wf1._shapes()[0][2] == 'res_net_1d/res_block_1d/first_conv/conv_1d/w' and len(wf1._shapes()[1]) == 10
  # the two convolutional types with kernels: constructor arguments and Sonnet variable names
  hp.set_hparam('wavefunction_type', 'conv_2d')
  hp.set_hparam('size_x', 4); hp.set_hparam('size_y', 6); hp.set_hparam('num_conv_layers', 3)
  wf = wavefunctions.build_wavefunction(hp)
  assert isinstance(wf, wavefunctions.Conv2DNetwork)
  wf._n_sites = 24
  names, shapes = wf._shapes()
  assert names[:4] == ['conv_2d_network/conv_2d_periodic/conv_2d/w', 'conv_2d_network/conv_2d_periodic/conv_2d/b',
                       'conv_2d_network/conv_2d_periodic_1/conv_2d/w', 'conv_2d_network/conv_2d_periodic_1/conv_2d/b']
  assert shapes == [(5, 5, 1, 16), (16,), (5, 5, 16, 16), (16,), (5, 5, 16, 16), (16,)]
  assert wf._engine_spec() == dict(ansatz='conv_2d', num_layers=3, layer_size=16, nonlinearity='relu',
                                   output_activation='exp', kernel_size=5, size_x=4, size_y=6)
  hp.set_hparam('wavefunction_type', 'res_net_2d')
  wf = wavefunctions.build_wavefunction(hp)
  wf._n_sites = 24
  names, shapes = wf._shapes()
  assert names[2] == 'res_net_2d/res_block_2d/first_conv/conv_2d/w'
  assert names[-1] == 'res_net_2d/res_block_2d_1/second_conv/conv_2d/b' and len(shapes) == 10
  import copy as _copy
  twin = _copy.deepcopy(wf)
  assert isinstance(twin, wavefunctions.ResNet2D) and twin._unique_name == 'dc_res_net_2d'
  assert twin._engine_spec() == wf._engine_spec()
  hp.set_hparam('conv_strides', 2)
  with pytest.raises(ValueError):
    wavefunctions.build_wavefunction(hp)
  with pytest.raises(AttributeError):   # defect B1 of the reference: ITSWO cannot be built
    training.GROUND_STATE_OPTIMIZERS['ITSWO']().build_opt_ops(None, None, hp, {})
  assert training.TrainOpsTraditional._fields == (
      'accumulate_gradients', 'apply_gradients', 'reset_gradients', 'mc_step', 'acc_rate',
      'metrics', 'epoch_increment', 'update_wf_norm')
  assert training.TrainOpsSWO._fields == (
      'train_step', 'accumulate_gradients', 'apply_gradients', 'reset_gradients', 'mc_step',
      'acc_rate', 'metrics', 'energy', 'update_supervisor', 'update_normalization',
      'epoch_increment', 'update_wf_norm')
  assert evaluation.EvalOps._fields == ('value', 'mc_step', 'acceptance_rate',
                                        'placeholder_input', 'wavefunction_value')
  assert [r.value for r in graph_builders.ResourceName] == [
      'CONFIGS', 'TARGET_CONFIGS', 'TARGET_PSI', 'TRAINING_PSI', 'MONTE_CARLO_SAMPLING']


def test_wavefunction_variables_names_order_and_deepcopy():
  hp = utils.create_hparams(wavefunction_type='fully_connected', num_fc_layers=2, fc_layer_size=4)
  wf = wavefunctions.build_wavefunction(hp)
  assert wf._unique_name == 'fully_connected_network'
  with pytest.raises(ValueError):
    wf.get_trainable_variables()             # sonnet creates variables at first connection
  wf._n_sites = 3
  wf.initialize(seed=0)
  names = [v.name for v in wf.get_trainable_variables()]
  assert names == ['fully_connected_network/linear/w', 'fully_connected_network/linear/b',
                   'fully_connected_network/linear_1/w', 'fully_connected_network/linear_1/b',
                   'fully_connected_network/linear_2/w', 'fully_connected_network/linear_2/b']
  shapes = [v.shape for v in wf.get_trainable_variables()]
  assert shapes == [(3, 4), (4,), (4, 4), (4,), (4, 1), (1,)]
  assert wf.num_params == vo.num_params(3, 4, 2)
  theta = wf._get_theta()
  layers_ = vo.unpack(theta, 3, 4, 2)
  np.testing.assert_array_equal(wf.get_trainable_variables()[2].eval(), layers_[1][0])
  assert np.abs(layers_[0][0]).max() <= 2 / np.sqrt(3) + 1e-6 and (layers_[0][1] == 0).all()
  import copy
  twin = copy.deepcopy(wf)
  assert twin._unique_name == 'dc_fully_connected_network'
  assert (twin._num_layers, twin._layer_size) == (2, 4) and twin._theta is None
  assert wf._exp_norm_shift == np.float32(-10.0)
  # second instance gets a uniquified scope
  assert wavefunctions.build_wavefunction(hp)._unique_name == 'fully_connected_network_1'


def test_saver_roundtrip_and_latest_checkpoint(tmp_path):
  hp = utils.create_hparams(wavefunction_type='fully_connected', num_fc_layers=1, fc_layer_size=4)
  wf = wavefunctions.build_wavefunction(hp)
  wf._n_sites = 3
  wf.initialize(seed=1)
  sess = session.Session()
  saver = session.Saver(wf.get_trainable_variables(), max_to_keep=2)
  d = str(tmp_path)
  assert session.latest_checkpoint(d) is None
  for e in range(3):
    saver.save(sess, os.path.join(d, 'model_prior_{}_epochs'.format(e)))
  assert not os.path.exists(os.path.join(d, 'model_prior_0_epochs.npz'))   # max_to_keep
  latest = session.latest_checkpoint(d)
  assert latest.endswith('model_prior_2_epochs')
  theta = wf._get_theta().copy()
  wf.initialize(seed=2)
  assert not np.array_equal(wf._get_theta(), theta)
  saver.restore(sess, latest)
  np.testing.assert_array_equal(wf._get_theta(), theta)


def test_session_runs_handles():
  sess = session.Session()
  log = []
  a = session.Op(lambda: log.append('a'), 'a')
  b = session.Tensor(lambda: 3.5, 'b')
  assert sess.run(b) == 3.5
  assert sess.run([a, b, None]) == [None, 3.5, None]
  sess.run(session.group(a, [a, None]))
  assert log == ['a', 'a', 'a']
  with pytest.raises(TypeError):
    sess.run(42)


def test_lattice_and_j_file(tmp_path):
  assert lattice.chain_bonds(4) == [(0, 1), (1, 2), (2, 3), (3, 0)]
  t = lattice.torus_bonds(4, 4)
  assert len(t) == 32 and len(set(map(frozenset, t))) == 32
  assert lattice.torus_bonds(10, 10) == vo.torus_bonds(10, 10)
  assert len(lattice.torus_bonds(16, 16, True)) == 1024          # BASELINE config 5
  assert lattice.load_bonds(str(tmp_path), 5) == lattice.chain_bonds(5)
  lattice.write_bonds(str(tmp_path), t)
  assert lattice.load_bonds(str(tmp_path), 16) == [list(b) for b in t]


def test_get_configs_contract_without_gpu():
  shared = {}
  cfg = graph_builders.get_configs(shared, 8, 6)
  assert graph_builders.ResourceName.CONFIGS in shared
  assert graph_builders.get_configs(shared, 8, 6) is cfg
  with pytest.raises(ValueError, match='Size of existing variable does not match'):
    graph_builders.get_configs(shared, 9, 6)
  v = cfg.eval()
  assert v.shape == (8, 6) and (v.sum(1) == 0).all()
  other = graph_builders.get_configs(shared, 8, 6, include=False,
                                     configs_id=graph_builders.ResourceName.TARGET_CONFIGS)
  assert graph_builders.ResourceName.TARGET_CONFIGS not in shared and other is not cfg


def test_hamiltonian_holds_per_bond_couplings():
  h = operators.HeisenbergHamiltonian([(0, 1), [1, 2]], -1.0, 1.0)
  assert h._bonds_list == [(0, 1), (1, 2)]
  np.testing.assert_array_equal(h._j_x, [-1, -1]); np.testing.assert_array_equal(h._j_z, [1, 1])


def test_tf_v2_checkpoint_bundle_round_trip(tmp_path):
  """cgs_vmc_amd/tf_checkpoint.py: CRC32C known answer, table framing invariants, and a
  writer -> reader round trip with the variable names a reference checkpoint carries."""
  import struct
  from cgs_vmc_amd import tf_checkpoint as tfc
  assert tfc.crc32c(b'123456789') == 0xE3069283          # RFC 3720 / Castagnoli check value
  assert tfc.masked_crc32c(b'') == (((0 >> 15) | (0 << 17)) + 0xa282ead8) & 0xFFFFFFFF
  rng = np.random.default_rng(0)
  tensors = {
      'fully_connected_network/linear/w': rng.standard_normal((16, 32)).astype(np.float32),
      'fully_connected_network/linear/b': np.zeros(32, np.float32),
      'fully_connected_network/linear_1/w': rng.standard_normal((32, 1)).astype(np.float32),
      'fully_connected_network/linear_1/b': rng.standard_normal(1).astype(np.float32),
      'num_epochs': np.array(17, np.int32),                  # scalar: empty shape
      'beta1_power': np.array(0.5, np.float64),
  }
  for i in range(40):                                        # more than one restart interval
    tensors['extra/v_%02d' % i] = rng.integers(-5, 5, (3, i + 1)).astype(np.int64)
  prefix = str(tmp_path / 'model_prior_3_epochs')
  tfc.write_bundle(prefix, tensors)
  assert tfc.bundle_exists(prefix) and os.path.exists(prefix + '.data-00000-of-00001')
  raw = open(prefix + '.index', 'rb').read()
  assert struct.unpack('<Q', raw[-8:])[0] == 0xdb4775248b80fb57 and len(raw) > 48
  back = tfc.read_bundle(prefix)
  assert set(back) == set(tensors)
  for k, v in tensors.items():
    assert back[k].dtype == v.dtype and back[k].shape == v.shape
    np.testing.assert_array_equal(back[k], v)
  # corruption is detected
  data = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
  data[5] ^= 0xFF
  open(prefix + '.data-00000-of-00001', 'wb').write(bytes(data))
  with pytest.raises(ValueError, match='CRC32C'):
    tfc.read_bundle(prefix)
  assert tfc.read_bundle(prefix, verify=False).keys() == tensors.keys()


def test_saver_restores_a_tf_bundle(tmp_path, monkeypatch):
  """Saver.restore / latest_checkpoint accept the reference's V2 bundle next to the .npz form."""
  from cgs_vmc_amd import session as session_lib
  vals = {'a/w': np.arange(6, dtype=np.float32).reshape(2, 3), 'a/b': np.ones(3, np.float32)}
  store = {k: np.zeros_like(v) for k, v in vals.items()}
  def var(name):
    return session_lib.Variable(name, vals[name].shape, lambda: store[name],
                                lambda value: store.__setitem__(name, np.asarray(value, np.float32).reshape(vals[name].shape)),
                                trainable=True)
  variables = [var('a/w'), var('a/b')]
  for k in vals:
    store[k] = vals[k].copy()
  monkeypatch.setenv('CGS_VMC_CHECKPOINT_FORMAT', 'tf')
  saver = session_lib.Saver(variables)
  path = saver.save(None, str(tmp_path / 'model_prior_0_epochs'))
  assert os.path.exists(path + '.index') and not os.path.exists(path + '.npz')
  assert session_lib.latest_checkpoint(str(tmp_path)) == path
  for k in vals:
    store[k] = np.zeros_like(vals[k])
  saver.restore(None, path)
  for k in vals:
    np.testing.assert_array_equal(store[k], vals[k])


def test_heisenberg_bond_is_the_one_bond_hamiltonian():
  """operators.HeisenbergBond (operators.py:128-135): constructor signature and bond bookkeeping
  (its kernels are covered by tests/test_gpu_api.py)."""
  b = operators.HeisenbergBond((3, 5), 0.5, 2.0)
  assert isinstance(b, operators.Operator) and b._bond == (3, 5)
  assert b._bonds_list == [(3, 5)] and b._j_x.tolist() == [0.5] and b._j_z.tolist() == [2.0]


def test_bench_refuses_a_rank_count_that_contradicts_the_environment():
  """`--gpus N` must equal WORLD_SIZE when ranks already exist (a mislabelled
  single-rank run); checked before anything touches the GPU, so it runs on CPU."""
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, WORLD_SIZE='1', RANK='0')
  p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '4'], env=env,
                     stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
  assert p.returncode == 2 and b'WORLD_SIZE=1' in p.stderr and not p.stdout.strip()
