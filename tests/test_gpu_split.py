"""GPU: the 3 x bf16 split experiment of the row kernel (CGS_VMC_SPLIT_BF16=1, cgs_vmc_amd/csrc/tail_split.hip:
fp32 results from the bf16 matrix cores -- every f32 operand as three bf16 terms, six products accumulated
in fp32) at the SAME tolerances as the native fp32 kernel's logit and local-energy cases
(tests/test_gpu_engine.py, tests/test_gpu_fullsize.py), with its maximum error against the fp64 oracle
reported next to the native kernel's on the same inputs."""
import numpy as np
import pytest

from oracle import vmc_oracle as vo

pytestmark = pytest.mark.gpu

# (n, h, L, b, lattice): padded to 256 units in every case (the experiment covers 193 .. 256 units)
SHAPES = [(16, 256, 2, 96, 'torus4x4'), (36, 200, 3, 70, 'torus6x6'), (100, 256, 3, 160, 'torus10x10'),
          (40, 256, 4, 33, 'chain')]


def _bonds(kind, n):
  if kind == 'chain':
    return vo.chain_bonds(n)
  side = int(kind[5:].split('x')[0])
  return vo.torus_bonds(side, n // side)


def _engine(monkeypatch, split, n, h, L, b, kind, seed=0):
  from cgs_vmc_amd.engine import VmcEngine
  if split:
    monkeypatch.setenv('CGS_VMC_SPLIT_BF16', str(int(split)))       # True / 1: row kernel; 2: row kernel + sampler
  else:
    monkeypatch.delenv('CGS_VMC_SPLIT_BF16', raising=False)
  rng = np.random.default_rng(seed)
  theta = vo.init_params(n, h, L, rng) + (0.03 * rng.standard_normal(vo.num_params(n, h, L))).astype(np.float32)
  cfg = vo.random_configurations(n, b, np.random.RandomState(seed + 1))
  bonds = _bonds(kind, n)
  eng = VmcEngine(n, b, L, h, seed=2024)
  eng.set_params(theta); eng.set_configs(cfg); eng.set_bonds(bonds, -1.0, 1.0)
  assert eng.kernel_path() == ({1: 4, 2: 5}[int(split)] if split else 0)
  return eng, theta, cfg, bonds


@pytest.mark.parametrize('n,h,L,b,kind', SHAPES)
def test_split_rows_meet_the_native_tolerances(monkeypatch, n, h, L, b, kind):
  ref_logit = ref_eloc = None
  errs = {}
  for split in (False, True):
    eng, theta, cfg, bonds = _engine(monkeypatch, split, n, h, L, b, kind)
    if ref_logit is None:
      amp = lambda c: vo.fc_psi(theta, c, h, L, dtype=np.float64)
      ref_logit = vo.fc_logit(theta, cfg, h, L, dtype=np.float64)
      ref_eloc = vo.local_value(amp, cfg, bonds, -1.0, 1.0, dtype=np.float64)
    logit = eng.amplitude()[0].astype(np.float64)
    ext = eng.amplitude(cfg[:9])[0].astype(np.float64)          # rows supplied by the caller
    eloc = eng.local_energy()[0].astype(np.float64)
    e_logit = np.abs(logit - ref_logit).max() / max(1.0, np.abs(ref_logit).max())
    e_eloc = np.abs(eloc - ref_eloc).max() / max(1.0, np.abs(ref_eloc).max())
    errs[split] = (e_logit, e_eloc)
    assert e_logit <= 2e-5 and e_eloc <= 2e-4, (split, errs)                       # tests/test_gpu_engine.py's bars
    assert np.abs(ext - ref_logit[:9]).max() <= 2e-5 * max(1.0, np.abs(ref_logit).max())
    eng.close()
  print('\nmax error / scale vs the fp64 oracle  logit: native {:.2e} split {:.2e}   E_loc: native {:.2e} split {:.2e}'
        .format(errs[False][0], errs[True][0], errs[False][1], errs[True][1]))
  # the split is an fp32-grade computation, not a bf16-grade one
  assert errs[True][0] <= 4 * errs[False][0] + 2e-6 and errs[True][1] <= 4 * errs[False][1] + 2e-5


def test_split_rows_at_config3_full_size(monkeypatch):
  """BASELINE config 3 (10x10, FC 3x256, 4096 chains) on the split kernel: logits and local energies of
  every 64th chain against the fp64 oracle before and after a sweep (tests/test_gpu_fullsize.py's bars);
  the sampler is the native fp32 kernel either way, so the chains are the native ones."""
  import bench
  from cgs_vmc_amd.engine import VmcEngine
  n, h, L, b = 100, 256, 3, 4096
  theta, cfg = bench.make_inputs(n, h, L, b, 0)
  bonds = vo.torus_bonds(10, 10, False)
  outs = {}
  for split in (False, True):
    if split:
      monkeypatch.setenv('CGS_VMC_SPLIT_BF16', '1')
    else:
      monkeypatch.delenv('CGS_VMC_SPLIT_BF16', raising=False)
    eng = VmcEngine(n, b, L, h, seed=2024)
    eng.set_params(theta); eng.set_configs(cfg); eng.set_bonds(bonds, -1.0, 1.0)
    assert eng.kernel_path() == (4 if split else 0)
    e0 = eng.local_energy()[0]
    eng.mc_steps(n)
    outs[split] = (e0, eng.get_configs(), eng.amplitude()[0], eng.local_energy()[0])
    eng.close()
  np.testing.assert_array_equal(outs[False][1], outs[True][1])            # same sampler, same chains
  pick = np.arange(0, b, 64)
  amp = lambda c: vo.fc_psi(theta, c, h, L, dtype=np.float64)
  moved = outs[True][1]
  ref_logit = vo.fc_logit(theta, moved[pick], h, L, dtype=np.float64)
  ref_e0 = vo.local_value(amp, cfg[pick], bonds, -1.0, 1.0, dtype=np.float64)
  ref_e1 = vo.local_value(amp, moved[pick], bonds, -1.0, 1.0, dtype=np.float64)
  for split in (False, True):
    e0, _, logit, e1 = outs[split]
    assert np.abs(e0[pick] - ref_e0).max() <= 2e-4 * max(1.0, np.abs(ref_e0).max()), split
    assert np.abs(e1[pick] - ref_e1).max() <= 2e-4 * max(1.0, np.abs(ref_e1).max()), split
  # (the cached logits after a sweep come from the sampler, not from the row kernel)
  assert np.abs(outs[True][2][pick] - ref_logit).max() <= 2e-5 * max(1.0, np.abs(ref_logit).max())
  assert np.abs(outs[True][3] - outs[False][3]).max() <= 2e-4 * max(1.0, np.abs(outs[False][3]).max())


@pytest.mark.parametrize('n,h,L,b,kind', SHAPES + [(100, 256, 6, 300, 'torus10x10')])
def test_ring_kernel_gives_the_bits_of_the_per_wave_stream(monkeypatch, n, h, L, b, kind):
  """Round 5: k_tail16r (weights once per workgroup through an LDS-DMA ring) against k_tail16s (every wave
  streams them itself, CGS_VMC_SPLIT_RING=0): the same products in the same order -- logits, rows supplied by the
  caller (ragged last tile) and local energies are the same bits, launch after launch."""
  outs = {}
  for ring in ('0', '1'):
    monkeypatch.setenv('CGS_VMC_SPLIT_RING', ring)
    eng, theta, cfg, bonds = _engine(monkeypatch, True, n, h, L, b, kind)
    res = []
    for _ in range(3 if ring == '1' else 1):
      res.append((eng.amplitude()[0], eng.amplitude(cfg[:37])[0], eng.local_energy()[0]))
      eng.mc_steps(n // 2, want_accepted=False)
    outs[ring] = res
    eng.close()
  # the sampler is the native kernel either way: the chains after each half sweep are the same, so launch k of
  # the ring engine is compared with ... the per-wave engine only ran launch 0; the later launches of the ring
  # engine are checked for finiteness and against the native tolerance below
  for a, r in zip(outs['0'][0], outs['1'][0]):
    np.testing.assert_array_equal(a, r)
  for res in outs['1'][1:]:
    assert all(np.all(np.isfinite(x)) for x in res)


def test_ring_kernel_at_config3_full_size_matches_the_per_wave_stream(monkeypatch):
  import bench
  from cgs_vmc_amd.engine import VmcEngine
  n, h, L, b = 100, 256, 3, 4096
  theta, cfg = bench.make_inputs(n, h, L, b, 0)
  bonds = vo.torus_bonds(10, 10, False)
  monkeypatch.setenv('CGS_VMC_SPLIT_BF16', '1')
  outs = {}
  for ring in ('0', '1'):
    monkeypatch.setenv('CGS_VMC_SPLIT_RING', ring)
    eng = VmcEngine(n, b, L, h, seed=2024)
    eng.set_params(theta); eng.set_configs(cfg); eng.set_bonds(bonds, -1.0, 1.0)
    assert eng.kernel_path() == 4
    e0 = eng.local_energy()[0]
    eng.mc_steps(n, want_accepted=False)
    e1 = eng.local_energy()[0]
    e1b = eng.local_energy()[0]                  # the same launch again: the same bits
    np.testing.assert_array_equal(e1, e1b)
    outs[ring] = (e0, e1, eng.amplitude(cfg[:1000])[0])
    eng.close()
  for a, r in zip(outs['0'], outs['1']):
    np.testing.assert_array_equal(a, r)


def test_split_is_refused_silently_where_it_does_not_apply(monkeypatch):
  """The switch only acts on shapes the experiment covers; everything else keeps its native kernels."""
  from cgs_vmc_amd.engine import VmcEngine
  monkeypatch.setenv('CGS_VMC_SPLIT_BF16', '1')
  for kw in (dict(layer_size=128), dict(layer_size=256, nonlinearity='tanh'), dict(layer_size=256, ansatz='rbm')):
    eng = VmcEngine(16, 32, 2, kw.pop('layer_size'), **kw)
    assert eng.kernel_path() == 0
    eng.close()


# --------------------------------------------------------------------------------------------- the split SAMPLER
# k_sweep16s (CGS_VMC_SPLIT_BF16=2): tests/test_gpu_engine.py's sampler cases at their tolerances, on 256-unit shapes
SAMPLER_SHAPES = [(16, 256, 2, 64, 'torus4x4'), (36, 200, 3, 70, 'torus6x6'), (100, 256, 3, 96, 'torus10x10'),
                  (40, 256, 4, 33, 'chain'),
                  (144, 256, 3, 40, 'torus12x12'), (256, 256, 6, 24, 'torus16x16')]   # 129 .. 256 sites: four Philox blocks per lane


@pytest.mark.parametrize('n,h,L,b,kind', SAMPLER_SHAPES)
def test_split_sampler_injected_step_matches_oracle(monkeypatch, n, h, L, b, kind):
  """test_gpu_engine.py::test_injected_mc_step_matches_oracle on the split sampler: accept masks bit-exact outside
  BASELINE.md's band, the chains exactly the accepted exchanges, the written-back cache = the new chains' logits."""
  eng, theta, cfg, _ = _engine(monkeypatch, 2, n, h, L, b, kind)
  amp = lambda c: vo.fc_psi(theta, c, h, L, dtype=np.float64)
  cur = cfg
  for step in range(6):
    u_sites, u_acc = vo.step_uniforms(99, np.arange(b), step, n)
    i_up, i_dn = vo.propose_exchange(cur, u_sites)
    new_ref, acc_ref, ratios = vo.mc_step(amp, cur, i_up, i_dn, u_acc)
    mask = eng.mc_step_injected(i_up, i_dn, u_acc)
    band = np.abs(ratios - np.sqrt(u_acc.astype(np.float64))) < 1e-4 * np.maximum(ratios, 1e-30)
    assert np.array_equal(mask[~band], acc_ref[~band])
    got = eng.get_configs()
    expect = cur.copy()
    rows = np.arange(b)[mask]
    expect[rows, i_dn[mask]] = 1.0
    expect[rows, i_up[mask]] = -1.0
    np.testing.assert_array_equal(got, expect)
    cur = got
    ref = vo.fc_logit(theta, cur, h, L, dtype=np.float64)
    assert np.abs(eng.amplitude()[0] - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
  eng.close()


@pytest.mark.parametrize('n,h,L,b,kind', SAMPLER_SHAPES)
def test_split_sampler_trajectory_follows_oracle(monkeypatch, n, h, L, b, kind):
  """vmc_mc_steps on the split sampler reproduces the oracle's chains step by step (chains whose accept test falls
  in the band are excluded from then on); one launch of 12 steps == 12 launches of one; two engines agree bit for
  bit (run-to-run identity inside the mode); a shard walks the rows of the whole batch (shard invariance)."""
  from cgs_vmc_amd.engine import VmcEngine
  eng, theta, cfg, bonds = _engine(monkeypatch, 2, n, h, L, b, kind)
  amp = lambda c: vo.fc_psi(theta, c, h, L, dtype=np.float64)
  cur = cfg.copy()
  ok = np.ones(b, bool)
  for step in range(12):
    u_sites, u_acc = vo.step_uniforms(2024, np.arange(b), step, n)
    i_up, i_dn = vo.propose_exchange(cur, u_sites)
    cur, acc, ratios = vo.mc_step(amp, cur, i_up, i_dn, u_acc)
    ok &= ~(np.abs(ratios - np.sqrt(u_acc.astype(np.float64))) < 1e-4 * np.maximum(ratios, 1e-30))
    eng.mc_steps(1)
    np.testing.assert_array_equal(eng.get_configs()[ok], cur[ok])
  assert ok.sum() > b // 2
  eng2, _, _, _ = _engine(monkeypatch, 2, n, h, L, b, kind)
  eng2.mc_steps(12)
  np.testing.assert_array_equal(eng2.get_configs(), eng.get_configs())
  np.testing.assert_array_equal(eng2.amplitude()[0], eng.amplitude()[0])
  half = VmcEngine(n, b - b // 2, L, h, seed=2024, chain_offset=b // 2)
  half.set_params(theta); half.set_configs(cfg[b // 2:]); half.set_bonds(bonds, -1.0, 1.0)
  assert half.kernel_path() == 5
  half.mc_steps(12)
  np.testing.assert_array_equal(half.get_configs(), eng2.get_configs()[b // 2:])
  # the exact cache the sampler leaves and the local energies of its chains, against the fp64 oracle
  got = eng2.get_configs()
  ref = vo.fc_logit(theta, got, h, L, dtype=np.float64)
  assert np.abs(eng2.amplitude()[0] - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
  ref_e = vo.local_value(amp, got, bonds, -1.0, 1.0, dtype=np.float64)
  assert np.abs(eng2.local_energy()[0] - ref_e).max() <= 2e-4 * max(1.0, np.abs(ref_e).max())
  eng.close(); eng2.close(); half.close()


def test_split_sampler_at_config3_full_size(monkeypatch):
  """BASELINE config 3 on both split kernels: after a sweep the chains are the native sampler's except where an
  accept test fell into the band (a handful of 409,600 decisions), logits and local energies of every 64th chain
  within the native bars of the fp64 oracle, gradient accumulators within theirs of the native engine's."""
  import bench
  from cgs_vmc_amd import _hip
  from cgs_vmc_amd.engine import VmcEngine
  n, h, L, b = 100, 256, 3, 4096
  theta, cfg = bench.make_inputs(n, h, L, b, 0)
  bonds = vo.torus_bonds(10, 10, False)
  outs = {}
  for mode in (None, '2'):
    if mode:
      monkeypatch.setenv('CGS_VMC_SPLIT_BF16', mode)
    else:
      monkeypatch.delenv('CGS_VMC_SPLIT_BF16', raising=False)
    eng = VmcEngine(n, b, L, h, seed=2024)
    eng.set_params(theta); eng.set_configs(cfg); eng.set_bonds(bonds, -1.0, 1.0)
    assert eng.kernel_path() == (5 if mode else 0)
    eng.mc_steps(n)
    got = eng.get_configs()
    eng.reset_accumulators(); eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
    outs[mode] = (got, eng.amplitude()[0], eng.local_energy()[0], eng.get_accumulators())
    eng.close()
  same = (outs[None][0] == outs['2'][0]).all(1)
  assert same.mean() > 0.995, same.mean()          # band events only
  pick = np.arange(0, b, 64)
  amp = lambda c: vo.fc_psi(theta, c, h, L, dtype=np.float64)
  got = outs['2'][0]
  ref = vo.fc_logit(theta, got[pick], h, L, dtype=np.float64)
  assert np.abs(outs['2'][1][pick] - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
  ref_e = vo.local_value(amp, got[pick], bonds, -1.0, 1.0, dtype=np.float64)
  assert np.abs(outs['2'][2][pick] - ref_e).max() <= 2e-4 * max(1.0, np.abs(ref_e).max())
  if same.all():
    a0, a1 = outs[None][3], outs['2'][3]
    assert np.abs(a0 - a1).max() <= 2e-3 * np.abs(a0).max() + 1e-4


def test_split_adversarial_operands(monkeypatch):
  """Round 6: a bound under the experiment on operands chosen to hurt -- weight rows spanning 2^20
  in magnitude, hidden units in exactly cancelling pairs, the weights after some training epochs -- through
  tools/split_adversarial.py: both split kernels (CGS_VMC_SPLIT_BF16=2) against the native kernels and the fp64 oracle.
  The split's error is the native kernels' error (measured worst ratio 1.06, profiles/r6_split_adversarial.txt; bound
  here: 2), and no Metropolis decision differs from the fp64 decision outside the band the native kernel is allowed."""
  import importlib.util
  import os
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  spec = importlib.util.spec_from_file_location('split_adversarial', os.path.join(root, 'tools', 'split_adversarial.py'))
  tool = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(tool)
  saved = os.environ.get('CGS_VMC_SPLIT_BF16')
  try:
    rng = np.random.default_rng(1)
    cases = {'range': tool.make_range(rng), 'cancel': tool.make_cancel(rng), 'trained': tool.make_trained(12)[0]}
    for name, theta in cases.items():
      nat, ln, en = tool.evaluate(theta, False)
      spl, ls, es = tool.evaluate(theta, True)
      assert spl['logit_err'] <= 2e-5 and spl['eloc_err'] <= 2e-4, (name, spl)           # the native kernels' bars
      assert spl['logit_err'] <= 2 * max(nat['logit_err'], 1e-7) and spl['eloc_err'] <= 2 * max(nat['eloc_err'], 1e-6), (name, nat, spl)
      assert spl['wrong_decisions'] == 0 and nat['wrong_decisions'] == 0 and spl['decisions'] >= 700, (name, nat, spl)
  finally:
    if saved is None:
      os.environ.pop('CGS_VMC_SPLIT_BF16', None)
    else:
      os.environ['CGS_VMC_SPLIT_BF16'] = saved
