"""GPU: the eight-chain sampler k_sweep8 (cgs_vmc_amd/csrc/sweep8.hip) produces the chains of k_sweep16 BIT FOR BIT.

Both kernels implement graph_builders.py:38-89; k_sweep8 serves batches that would leave sixteen-chain tiles on at most
half of the CUs (BASELINE configs 2 and 5: 1,024 chains per GPU).  It runs the H x H layers on the 4x4x1 MFMA shape in
the k order of the 16x16x4 chain (tools/ubench/mfma_order.hip: that form is a chain of fused multiply-adds in slot
order) and repeats k_sweep16's summation trees everywhere else, so every Metropolis decision is the same decision:
no tolerance below -- chains, accept counts, the z1 / logit cache (through the local energies it feeds), the bond
census and the activations handed to the gradient path (through the gradient sums) are compared with array_equal.
The oracle trajectories of tests/test_gpu_engine.py run on whichever kernel vmc_create picks (k_sweep8 for most of
their small batches); one of them is repeated here with the tile forced both ways."""
import numpy as np
import pytest

from oracle import vmc_oracle as vo

pytestmark = pytest.mark.gpu

SHAPES = [
    # n_sites, H, L, B, lattice
    (36, 128, 3, 1024, 'torus6x6'),    # BASELINE config 2 at full size: 128 eight-chain tiles, W1 in LDS, all weights resident
    (36, 128, 3, 203, 'torus6x6'),     # ragged last tile
    (100, 256, 3, 96, 'torus10x10'),   # config 3's ansatz: W1 in LDS, layer 1 streamed
    (256, 256, 6, 40, 'j1j2_16x16'),   # config 5's ansatz: every lane owns a site block, W1 from L2, 1,024 bonds
    (37, 100, 4, 50, 'chain'),         # n_sites not a multiple of 4, units padded to 128, three H x H layers at 128 units
    (128, 128, 2, 24, 'chain'),        # n_sites = units = 128: the second Philox call for the acceptance uniform
    (20, 256, 2, 9, 'chain'),          # one H x H layer at 256 units: nothing streams
]


def _bonds(kind, n):
  if kind == 'chain':
    return vo.chain_bonds(n), -1.0
  if kind == 'j1j2_16x16':
    nn = vo.torus_bonds(16, 16)
    idx = lambda x, y: (x % 16) * 16 + (y % 16)
    nnn = [(idx(x, y), idx(x + 1, y + 1)) for x in range(16) for y in range(16)] + \
          [(idx(x, y), idx(x + 1, y - 1)) for x in range(16) for y in range(16)]
    return list(nn) + nnn, -1.0
  lx = int(kind[5:].split('x')[0])
  return vo.torus_bonds(lx, n // lx), -1.0


def _run(eng, cfg, tile, n, steps_a, steps_b):
  """two launches (the second starts from the first one's cache), then everything the sampler hands on"""
  from cgs_vmc_amd import _hip
  assert eng.sweep_tile(tile) == tile
  eng.set_configs(cfg)
  eng.step_counter = 0
  out = {}
  out['acc_a'] = eng.mc_steps(steps_a)
  out['cfg_a'] = eng.get_configs()
  out['eloc_a'] = eng.local_energy()[0]          # z1 / logit cache + census of launch a
  eng.reset_accumulators()
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)   # marks the activations as wanted by the next launch
  out['acc_b'] = eng.mc_steps(steps_b)            # cache_in_valid launch, hands activations on
  out['cfg_b'] = eng.get_configs()
  out['logit_b'] = eng.amplitude()[0]
  eng.reset_accumulators()
  eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
  out['eloc_b'] = eng.local_energy()[0]
  out['acc_vec'] = eng.get_accumulators()
  return out


@pytest.mark.parametrize('n,h,L,b,kind', SHAPES)
def test_eight_chain_tiles_give_the_chains_of_sixteen_chain_tiles(n, h, L, b, kind):
  from cgs_vmc_amd.engine import VmcEngine
  rng = np.random.default_rng(5)
  theta = vo.init_params(n, h, L, rng)
  theta += (0.05 * rng.standard_normal(theta.size)).astype(np.float32)
  cfg = vo.random_configurations(n, b, np.random.RandomState(6))
  bonds, jx = _bonds(kind, n)
  eng = VmcEngine(n, b, L, h, seed=2024)
  eng.set_params(theta)
  eng.set_bonds(bonds, jx, 1.0)
  steps_a, steps_b = n + 3, 2 * n
  ref = _run(eng, cfg, 16, n, steps_a, steps_b)
  got = _run(eng, cfg, 8, n, steps_a, steps_b)
  assert 0 < ref['acc_a'] <= steps_a * b
  assert (ref['cfg_b'] != cfg).any()
  for k in ref:
    np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
  eng.close()


def test_vmc_create_picks_eight_chain_tiles_for_half_empty_chips(monkeypatch):
  from cgs_vmc_amd.engine import VmcEngine
  eng = VmcEngine(36, 1024, 3, 128)            # config 2: 64 sixteen-chain tiles on 256 CUs
  assert eng.sweep_tile() == 8
  eng.close()
  eng = VmcEngine(100, 4096, 3, 256)           # config 3: one sixteen-chain tile per CU
  assert eng.sweep_tile() == 16
  eng.close()
  eng = VmcEngine(36, 1024, 3, 128, nonlinearity='tanh')   # no k_sweep8 for this activation
  assert eng.sweep_tile() == 16
  with pytest.raises(Exception):
    eng.sweep_tile(8)
  eng.close()
  monkeypatch.setenv('CGS_VMC_SWEEP_TILE', '16')
  eng = VmcEngine(36, 1024, 3, 128)
  assert eng.sweep_tile() == 16
  eng.close()


@pytest.mark.parametrize('tile', [8, 16])
def test_trajectory_follows_oracle_on_either_tile(tile):
  """tests/test_gpu_engine.py::test_sampler_trajectory_follows_oracle with the tile forced"""
  from cgs_vmc_amd.engine import VmcEngine
  n, h, L, b = 252, 256, 3, 20
  rng = np.random.default_rng(0)
  theta = vo.init_params(n, h, L, rng)
  theta += (0.05 * rng.standard_normal(theta.size)).astype(np.float32)
  cfg = vo.random_configurations(n, b, np.random.RandomState(1))
  eng = VmcEngine(n, b, L, h, seed=2024)
  eng.set_params(theta); eng.set_configs(cfg); eng.set_bonds(vo.chain_bonds(n), -1.0, 1.0)
  assert eng.sweep_tile(tile) == tile
  amp = lambda c: vo.fc_psi(theta, c, h, L, dtype=np.float64)
  cur = cfg.copy()
  ok = np.ones(b, bool)
  for step in range(10):
    u_sites, u_acc = vo.step_uniforms(2024, np.arange(b), step, n)
    i_up, i_dn = vo.propose_exchange(cur, u_sites)
    cur, acc, ratios = vo.mc_step(amp, cur, i_up, i_dn, u_acc)
    ok &= ~(np.abs(ratios - np.sqrt(u_acc.astype(np.float64))) < 1e-4 * np.maximum(ratios, 1e-30))
    eng.mc_steps(1)
    np.testing.assert_array_equal(eng.get_configs()[ok], cur[ok])
  assert ok.sum() > b // 2
  eng.close()


def test_sharding_changes_the_tile_not_the_chains():
  """4,096 chains on one engine take sixteen-chain tiles, the same chains as four shards of 1,024 take eight-chain tiles
  (vmc_create's rule looks at the LOCAL batch): chains, accept counts and local energies are the same bits -- the
  property that keeps results independent of the number of GPUs (Philox keyed by the global chain id)."""
  from cgs_vmc_amd.engine import VmcEngine
  n, h, L, b, shards = 100, 256, 3, 4096, 4
  rng = np.random.default_rng(11)
  theta = vo.init_params(n, h, L, rng)
  theta += (0.05 * rng.standard_normal(theta.size)).astype(np.float32)
  cfg = vo.random_configurations(n, b, np.random.RandomState(12))
  bonds = vo.torus_bonds(10, 10)
  full = VmcEngine(n, b, L, h, seed=2024)
  full.set_params(theta); full.set_configs(cfg); full.set_bonds(bonds, -1.0, 1.0)
  assert full.sweep_tile() == 16
  acc_full = full.mc_steps(2 * n)
  cfg_full = full.get_configs()
  eloc_full = full.local_energy()[0]
  full.close()
  lb = b // shards
  acc_sum = 0
  for r in range(shards):
    eng = VmcEngine(n, lb, L, h, seed=2024, chain_offset=r * lb)
    eng.set_params(theta); eng.set_configs(cfg[r * lb:(r + 1) * lb]); eng.set_bonds(bonds, -1.0, 1.0)
    assert eng.sweep_tile() == 8
    acc_sum += eng.mc_steps(2 * n)
    np.testing.assert_array_equal(eng.get_configs(), cfg_full[r * lb:(r + 1) * lb])
    np.testing.assert_array_equal(eng.local_energy()[0], eloc_full[r * lb:(r + 1) * lb])
    eng.close()
  assert acc_sum == acc_full


@pytest.mark.parametrize('oact', ['tanh', 'identity', 'sigmoid'])
def test_non_exp_output_activations_on_eight_chain_tiles(oact):
  """psi = g(x) with g != exp: the accept rule is |g(x')| / |g(x)| > sqrt(u) in the linear domain (common.hpp:
  vmc_out_accept), evaluated by the group that owns the chain in k_sweep8 as by the owning wave in k_sweep16 -- the same
  chains, accept counts and local energies (tests/test_gpu_activations.py checks the sixteen-chain kernel against the
  oracle at 64 units, where no eight-chain kernel exists)."""
  from cgs_vmc_amd.engine import VmcEngine
  n, h, L, b = 16, 128, 3, 72
  rng = np.random.default_rng(21)
  theta = vo.init_params(n, h, L, rng)
  theta += (0.05 * rng.standard_normal(theta.size)).astype(np.float32)
  theta[-1] = 0.8                                   # b_out: keeps g(x) away from 0 for the bounded activations
  cfg = vo.random_configurations(n, b, np.random.RandomState(22))
  eng = VmcEngine(n, b, L, h, output_activation=oact, seed=2024)
  eng.set_params(theta); eng.set_bonds(vo.torus_bonds(4, 4), -1.0, 1.0)
  out = {}
  for tile in (16, 8):
    assert eng.sweep_tile(tile) == tile
    eng.set_configs(cfg)
    eng.step_counter = 0
    acc = eng.mc_steps(5 * n)
    out[tile] = (acc, eng.get_configs(), eng.amplitude()[0], eng.local_energy()[0])
  assert 0 < out[16][0] < 5 * n * b
  for a, bb in zip(out[16], out[8]):
    np.testing.assert_array_equal(a, bb)
  eng.close()


@pytest.mark.parametrize('optimizer', ['EnergyGradient', 'LogOverlapITSWO'])
def test_training_epochs_do_not_depend_on_the_tile(optimizer):
  """Two device-resident epochs (vmc_epoch_energy_gradient + Adam / vmc_epoch_log_overlap, whose supervisor refresh is a
  zero-step sampler launch) from the same start on sixteen- and on eight-chain tiles: the parameters, the Adam moments
  and the chains afterwards are the same bits (reference defaults' shape: 40 sites, 80 units padded to 128, 200 chains)."""
  from cgs_vmc_amd.engine import VmcEngine
  n, h, L, b = 40, 80, 3, 200
  rng = np.random.default_rng(31)
  theta = vo.init_params(n, h, L, rng)
  cfg = vo.random_configurations(n, b, np.random.RandomState(32))
  eng = VmcEngine(n, b, L, h, seed=2024)
  eng.set_bonds(vo.chain_bonds(n), -1.0, 1.0)
  out = {}
  for tile in (16, 8):
    assert eng.sweep_tile(tile) == tile
    eng.set_params(theta)
    eng.set_adam_state(np.zeros(theta.size, np.float32), np.zeros(theta.size, np.float32), 0)
    eng.set_shift(-10.0)
    eng.set_configs(cfg)
    eng.step_counter = 0
    energies = []
    for epoch in range(2):
      if optimizer == 'EnergyGradient':
        eng.epoch_energy_gradient(3 * n, 4, n, 1e10)
        energies.append(eng.apply_adam(0, 1e-3))
      else:
        eng.transfer_params()
        energies.append(eng.epoch_log_overlap(0.12, 3 * n, 4, n, 1e10, 1e-3, 0.9, 0.99, 1e-8))
    m, v, t = eng.get_adam_state()
    out[tile] = (np.float64(energies), eng.get_params(), m, v, eng.get_configs())
    assert t > 0 and np.isfinite(energies).all()
  assert (out[16][1] != theta).any()
  for a, bb in zip(out[16], out[8]):
    np.testing.assert_array_equal(a, bb)
  eng.close()
