"""GPU: the HIP path against the committed golden vectors (tests/golden/vmc_small.npz)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = ['chain16', 'torus4x4', 'torus6x6']


def _load(name):
  gold = np.load(os.path.join(HERE, 'golden', 'vmc_small.npz'))
  return {k.split('/', 1)[1]: gold[k] for k in gold.files if k.startswith(name + '/')}


def _engine(g):
  from cgs_vmc_amd.engine import VmcEngine
  n, h, L, b = [int(x) for x in g['shape']]
  eng = VmcEngine(n, b, L, h, seed=int(g['seed'][0]))
  eng.set_params(g['theta'])
  eng.set_configs(g['configs'])
  jx, jz, _ = g['couplings']
  eng.set_bonds(g['bonds'], jx, jz)
  return eng


@pytest.mark.parametrize('name', CASES)
def test_amplitudes_and_local_energy(name):
  g = _load(name)
  eng = _engine(g)
  logit, _ = eng.amplitude()
  assert np.abs(logit - g['logit']).max() < 2e-5 * max(1.0, np.abs(g['logit']).max())
  eloc, mean = eng.local_energy()
  assert np.abs(eloc - g['eloc']).max() < 2e-4 * max(1.0, np.abs(g['eloc']).max())
  diag, off = eng.local_energy_terms()
  np.testing.assert_allclose(diag, g['diag'], atol=1e-5)
  assert np.abs(off - g['offdiag_over_psi']).max() < 2e-4 * max(1.0, np.abs(g['offdiag_over_psi']).max())
  assert abs(mean - g['eloc'].mean()) < 2e-4 * max(1.0, abs(g['eloc'].mean()))
  eng.close()


@pytest.mark.parametrize('name', CASES)
def test_proposals_and_accepts(name):
  g = _load(name)
  for k, step in enumerate(g['steps']):
    eng = _engine(g)
    i_up, i_dn, u = eng.debug_proposals(int(step))
    np.testing.assert_array_equal(i_up, g['i_up'][k])          # integer work: bit-exact
    np.testing.assert_array_equal(i_dn, g['i_dn'][k])
    np.testing.assert_array_equal(u, g['u_accept'][k])
    mask = eng.mc_step_injected(g['i_up'][k], g['i_dn'][k], g['u_accept'][k])
    ratio = g['ratio'][k]
    band = np.abs(ratio - np.sqrt(g['u_accept'][k].astype(np.float64))) < 1e-4 * np.maximum(ratio, 1e-30)
    np.testing.assert_array_equal(mask[~band], g['accept'][k][~band])
    eng.close()


@pytest.mark.parametrize('name', CASES)
def test_gradient_accumulators_and_adam(name):
  from cgs_vmc_amd import _hip
  g = _load(name)
  eng = _engine(g)
  p = g['theta'].size
  eng.reset_accumulators()
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  acc = eng.get_accumulators()
  for got, ref in ((acc[:p], g['eg_g1']), (acc[p:2 * p], g['eg_g2'])):
    assert np.abs(got - ref).max() < 2e-3 * np.abs(ref).max() + 1e-4
  assert abs(acc[2 * p] - g['eg_scalars'][0]) < 2e-4 * max(1, abs(g['eg_scalars'][0]))
  assert acc[2 * p + 1] == g['eg_scalars'][1] and acc[2 * p + 4] == g['eg_scalars'][2]
  grad = eng.get_gradient(_hip.VMC_MODE_ENERGY_GRADIENT)
  assert np.abs(grad - g['eg_grad']).max() < 2e-3 * np.abs(g['eg_grad']).max() + 2e-4
  # the Adam step itself, from the golden gradient's accumulators (b_out excluded: its
  # gradient is identically zero and Adam turns rounding noise into +-lr, see test_gpu_api)
  eng.apply_adam(_hip.VMC_MODE_ENERGY_GRADIENT, 1e-3, 0.9, 0.99, 1e-8)
  th = eng.get_params()
  moved = np.abs(g['eg_grad']) > 50 * (2e-3 * np.abs(g['eg_grad']).max() + 2e-4)
  np.testing.assert_allclose(th[:-1][moved[:-1]], g['eg_theta_after_adam'][:-1][moved[:-1]], atol=2e-5)
  # LogOverlapITSWO accumulators
  eng.set_params(g['theta'])
  eng.set_params(g['theta_omega'], _hip.VMC_OMEGA)
  eng.set_shift(-9.0)
  eng.reset_accumulators()
  eng.accumulate(_hip.VMC_MODE_LOG_OVERLAP_ITSWO, float(g['couplings'][2]))
  acc = eng.get_accumulators()
  for got, ref in ((acc[:p], g['it_g1']), (acc[p:2 * p], g['it_g2'])):
    assert np.abs(got - ref).max() < 2e-3 * np.abs(ref).max() + 1e-4
  assert abs(acc[2 * p] - g['it_scalars'][0]) < 2e-4 * max(1, abs(g['it_scalars'][0]))
  assert abs(acc[2 * p + 2] - g['it_scalars'][2]) < 2e-4 * max(1, abs(g['it_scalars'][2]))
  grad = eng.get_gradient(_hip.VMC_MODE_LOG_OVERLAP_ITSWO)
  assert np.abs(grad - g['it_grad']).max() < 2e-3 * np.abs(g['it_grad']).max() + 2e-4
  eng.close()


@pytest.mark.parametrize('name', ['rbm_torus4x4', 'rbm_classic_chain12', 'rbm_classic_chain12_h400',
                                  'rbm_chain10_h260_l1'])
def test_rbm_golden(name):
  from cgs_vmc_amd.engine import VmcEngine
  gold = np.load(os.path.join(HERE, 'golden', 'rbm_wide.npz' if '_h' in name else 'rbm_small.npz'))
  g = {k.split('/', 1)[1]: gold[k] for k in gold.files if k.startswith(name + '/')}
  n, h, L, b = [int(x) for x in g['shape']]
  eng = VmcEngine(n, b, L, h, seed=int(g['seed'][0]), ansatz='rbm')
  eng.set_params(g['theta']); eng.set_configs(g['configs'])
  jx, jz, _ = g['couplings']
  eng.set_bonds(g['bonds'], jx, jz)
  logit, _ = eng.amplitude()
  assert np.abs(logit - g['logit']).max() < 2e-5 * max(1.0, np.abs(g['logit']).max())
  eloc, _ = eng.local_energy()
  assert np.abs(eloc - g['eloc']).max() < 2e-4 * max(1.0, np.abs(g['eloc']).max())
  i_up, i_dn, u = eng.debug_proposals(3)
  np.testing.assert_array_equal(i_up, g['i_up'])
  np.testing.assert_array_equal(i_dn, g['i_dn'])
  np.testing.assert_array_equal(u, g['u_accept'])
  eng.reset_accumulators()
  eng.accumulate(0)
  grad = eng.get_gradient(0)
  assert np.abs(grad - g['eg_grad']).max() < 2e-3 * np.abs(g['eg_grad']).max() + 2e-4
  mask = eng.mc_step_injected(g['i_up'], g['i_dn'], g['u_accept'])
  band = np.abs(g['ratio'] - np.sqrt(g['u_accept'].astype(np.float64))) < 1e-4 * np.maximum(g['ratio'], 1e-30)
  np.testing.assert_array_equal(mask[~band], g['accept'][~band])
  eng.close()


_CONV_WIDE = {'conv2d_4x4_f32': 'relu', 'conv2d_6x4_f24_cos': 'cos', 'resnet2d_4x4_f32': 'relu',
              'conv1d_12_f20_even': 'tanh'}


@pytest.mark.parametrize('name', ['conv2d_4x4', 'conv2d_6x4_even', 'resnet2d_4x4', 'conv1d_12_even', 'resnet1d_12']
                         + sorted(_CONV_WIDE))
def test_conv_golden(name):
  """Convolutional ansatz types against tests/golden/conv_small.npz / conv_wide.npz (tolerances of
  tests/test_gpu_conv.py: logits on the fp32 summation scale)."""
  from cgs_vmc_amd.engine import VmcEngine
  gold = np.load(os.path.join(HERE, 'golden', 'conv_wide.npz' if name in _CONV_WIDE else 'conv_small.npz'))
  g = {k.split('/', 1)[1]: gold[k] for k in gold.files if k.startswith(name + '/')}
  f, k, sx, sy, L, b = [int(x) for x in g['shape']]
  ansatz = {'conv2d': 'conv_2d', 'resnet2d': 'res_net_2d', 'conv1d': 'conv_1d', 'resnet1d': 'res_net_1d'}[name.split('_')[0]]
  nonlin = _CONV_WIDE.get(name, 'tanh' if name == 'conv2d_6x4_even' else 'relu')
  eng = VmcEngine(sx * sy, b, L, f, nonlinearity=nonlin, seed=int(g['seed'][0]), ansatz=ansatz, kernel_size=k,
                  size_x=sx, size_y=sy)
  eng.set_params(g['theta']); eng.set_configs(g['configs'])
  jx, jz, _ = g['couplings']
  eng.set_bonds(g['bonds'], jx, jz)
  logit, _ = eng.amplitude()
  assert (np.abs(logit - g['logit']) <= 1e-6 * g['scale'] + 2e-5).all()
  eloc, _ = eng.local_energy()
  assert np.abs(eloc - g['eloc']).max() < 2e-4 * max(1.0, np.abs(g['eloc']).max())
  i_up, i_dn, u = eng.debug_proposals(3)
  np.testing.assert_array_equal(i_up, g['i_up'])
  np.testing.assert_array_equal(i_dn, g['i_dn'])
  np.testing.assert_array_equal(u, g['u_accept'])
  eng.reset_accumulators()
  eng.accumulate(0)
  grad = eng.get_gradient(0)
  assert np.abs(grad - g['eg_grad']).max() < 2e-3 * np.abs(g['eg_grad']).max() + 2e-4
  mask = eng.mc_step_injected(g['i_up'], g['i_dn'], g['u_accept'])
  band = np.abs(g['ratio'] - np.sqrt(g['u_accept'].astype(np.float64))) < 1e-4 * np.maximum(g['ratio'], 1e-30)
  np.testing.assert_array_equal(mask[~band], g['accept'][~band])
  eng.close()
