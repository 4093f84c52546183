"""Diagnostic (GPU box): which (input tile, output tile) blocks of the split sampler's layer are wrong."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vmc_oracle as vo
os.environ['CGS_VMC_SPLIT_BF16'] = '2'
from cgs_vmc_amd.engine import VmcEngine
n, h, L, b = 16, 256, 2, 64
rng = np.random.default_rng(0)
base = vo.init_params(n, h, L, rng)
cfg = vo.random_configurations(n, b, np.random.RandomState(1))
eng = VmcEngine(n, b, L, h, seed=2024)
eng.set_configs(cfg); eng.set_bonds(vo.torus_bonds(4, 4), -1.0, 1.0)
def run(w2v, wov):
  theta = base.copy()
  (w1, b1), (w2, b2), (wo, bo) = vo.unpack(theta, n, h, L)
  w2[...] = w2v; wo[...] = wov[:, None]
  eng.set_params(theta); eng.set_configs(cfg)
  eng.transfer_params(); eng.accumulate(1, 0.12)
  w = eng.amplitude(which=1)[0]
  ref = vo.fc_logit(theta, cfg, h, L, dtype=np.float64)
  return np.abs(w - ref).max() / max(np.abs(ref).max(), 1e-9)
ones = np.ones(h, np.float32)
print('relative error of block (input tile s -> output tile t), W = 1/16 there, w_out = 1')
for s in (0, 7, 13, 14, 15):
  row = []
  for t in (0, 1, 6, 7, 8, 13, 14, 15):
    w2 = np.zeros((h, h), np.float32); w2[16 * s:16 * s + 16, 16 * t:16 * t + 16] = 1.0 / 16
    row.append('%.1e' % run(w2, ones))
  print('  in', s, '-> out (0,1,6,7,8,13,14,15):', ' '.join(row))
# per-chain pattern for one bad block
w2 = np.zeros((h, h), np.float32); w2[16 * 14:16 * 14 + 16, 0:16] = 1.0 / 16
theta = base.copy(); (w1, b1), (w2_, b2), (wo, bo) = vo.unpack(theta, n, h, L); w2_[...] = w2; wo[...] = 1.0
eng.set_params(theta); eng.set_configs(cfg); eng.transfer_params(); eng.accumulate(1, 0.12)
w = eng.amplitude(which=1)[0]; ref = vo.fc_logit(theta, cfg, h, L, dtype=np.float64)
print('in 14 -> out 0, per chain got/ref:', np.round(w[:16], 3), np.round(ref[:16], 3))
a1 = np.maximum(cfg @ w1 + b1, 0)
print('sum of input tile 14 / tile 15 / tile 13 activations per chain:', np.round(a1[:8, 224:240].sum(1), 3), np.round(a1[:8, 240:256].sum(1), 3), np.round(a1[:8, 208:224].sum(1), 3))
eng.close()
