"""Exact states for known-answer tests: Heisenberg ground states by exact diagonalisation (scipy) and a
fully_connected network that EQUALS one of them on its whole Sz = 0 sector.

Test infrastructure (CPU, fp64); nothing here is used by the product path.  The ED vector plays the
role FullVector plays in the reference (wavefunctions.py:1001-1055); the Hamiltonian is the standard
spin-1/2 one, sum_<ij> jz Sz_i Sz_j + jx/2 (S+_i S-_j + h.c.), i.e. operators.py:137-169's
1/4 jz s_i s_j on the diagonal and 1/2 jx between configurations that differ by one exchange."""
import itertools

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import vmc_oracle as vo


def ed_ground_state(n, bonds, jx, jz):
  """Exact ground state in the Sz=0 sector, basis = all +-1 configs with n/2 downs."""
  basis = [c for c in itertools.combinations(range(n), n // 2)]
  index = {c: k for k, c in enumerate(basis)}
  cfgs = np.ones((len(basis), n), np.float64)
  for k, c in enumerate(basis):
    cfgs[k, list(c)] = -1
  rows, cols, vals = [], [], []
  for k, c in enumerate(basis):
    down = set(c)
    d = 0.0
    for (i, j) in bonds:
      sz = cfgs[k, i] * cfgs[k, j]
      d += 0.25 * jz * sz
      if sz < 0:
        nd = set(down)
        if i in nd:
          nd.remove(i); nd.add(j)
        else:
          nd.remove(j); nd.add(i)
        rows.append(k); cols.append(index[tuple(sorted(nd))]); vals.append(0.5 * jx)
    rows.append(k); cols.append(k); vals.append(d)
  hmat = sp.csr_matrix((vals, (rows, cols)), shape=(len(basis),) * 2)
  w, v = spla.eigsh(hmat, k=1, which='SA')
  return w[0], v[:, 0], cfgs, index



def exact_fc_eigenstate(n, bonds, layer_size, num_layers, seed=7, jx=-1.0, jz=1.0):
  """FullyConnectedNetwork parameters (wavefunctions.py:331-371 layout, fp32) whose psi = exp(logit)
  is the exact ground state on every Sz = 0 configuration, for jx < 0 on a bipartite lattice (the
  Marshall-rotated sign, where the ground state is positive).

  The hidden layers keep their random initial weights; the last Linear(1) is solved so that
  logit(R) = log psi_ED(R) on all C(n, n/2) configurations, which needs layer_size >= C(n, n/2)
  (generic features are then linearly independent).  Returns (theta, E0, configs[dim, n], psi_ED)."""
  e0, vec, cfgs, _ = ed_ground_state(n, bonds, jx, jz)
  vec = vec * np.sign(vec[np.argmax(np.abs(vec))])
  assert (vec > 0).all(), 'ground state is not sign-free: use jx < 0 on a bipartite lattice'
  assert layer_size >= len(vec), 'need at least one hidden unit per basis configuration'
  theta = vo.init_params(n, layer_size, num_layers, np.random.default_rng(seed)).astype(np.float64)
  a = cfgs.copy()
  for (w, b) in vo.unpack(theta, n, layer_size, num_layers)[:-1]:
    a = np.maximum(a @ w + b, 0.0)
  target = np.log(vec)
  b_out = target.mean()
  w_out, _, rank, _ = np.linalg.lstsq(a, target - b_out, rcond=None)
  assert rank == len(vec)
  assert np.abs(a @ w_out + b_out - target).max() < 1e-9
  p = theta.size
  theta[p - 1 - layer_size:p - 1] = w_out
  theta[p - 1] = b_out
  return theta.astype(np.float32), float(e0), cfgs.astype(np.float32), vec


def exact_conv_eigenstate(ansatz, geom, num_layers, bonds, nonlinearity='relu', seed=11, jx=-1.0, jz=1.0):
  """Conv2DNetwork / Conv1DNetwork (and ResNet2D / ResNet1D: num_layers = blocks) parameters
  (wavefunctions.py:531-615 / 455-527 / 710-809) whose psi = exp(logit) is
  the exact Heisenberg ground state on the whole Sz = 0 sector of a small periodic lattice.

  The last convolution has no activation behind it and the logit is the sum of its output over sites and
  channels; on a torus every tap visits every site once, so
    logit(R) = sum_ci (sum_{tap, co} W_last[tap, ci, co]) S_ci(R) + N sum_co b_last[co],
    S_ci(R) = sum over sites of channel ci of the last convolution's INPUT.
  The earlier convolutions keep their random weights; W_last[tap, ci, 0] = w_ci / taps (other output channels
  zero) and b_last[0] = b / N with (w, b) the least-squares solution of sum_ci w_ci S_ci + b = log psi_ED.
  The features are translation invariant, like the ground state, so the system is consistent as soon as the
  filters outnumber the translation orbits of the sector (10 necklaces for 8 sites); the residual is asserted.
  Returns (theta, E0, configs[dim, n], psi_ED)."""
  f, k, sx, sy = geom
  n = sx * sy
  resnet = ansatz not in vo.CONV_PLAIN
  assert num_layers >= (1 if resnet else 2)
  e0, vec, cfgs, _ = ed_ground_state(n, bonds, jx, jz)
  vec = vec * np.sign(vec[np.argmax(np.abs(vec))])
  assert (vec > 0).all(), 'ground state is not sign-free: use jx < 0 on a bipartite lattice'
  theta = vo.conv_init_params(ansatz, geom, num_layers, np.random.default_rng(seed)).astype(np.float64)
  theta += 0.05 * np.random.default_rng(seed + 1).standard_normal(theta.size)      # non-zero biases
  _, tape, layers_ = vo.conv_forward(theta, cfgs, ansatz, geom, num_layers, nonlinearity, np.float64, return_tape=True)
  a_in = tape[-1][0]                                   # input of the last convolution: [dim, sx, sy, F]
  s_feat = a_in.reshape(a_in.shape[0], -1, a_in.shape[-1]).sum(1)                  # [dim, F]
  target = np.log(vec)
  if resnet:   # ResNet2D (wavefunctions.py:766-773): logit = sum(h + conv2(selu(conv1 h))); the shortcut's share
               # sum(h) is translation invariant as well and moves to the right-hand side
    h_prev = tape[-2][0]
    target = target - h_prev.reshape(h_prev.shape[0], -1).sum(1)
  design = np.concatenate([s_feat, np.ones((len(vec), 1))], 1)
  sol, _, _, _ = np.linalg.lstsq(design, target, rcond=None)
  assert np.abs(design @ sol - target).max() < 1e-8, 'too few filters for the translation orbits of this sector'
  w_ci, b = sol[:-1], sol[-1]
  shapes = vo.conv_param_shapes(ansatz, geom, num_layers)
  w_shape, b_shape = shapes[-2], shapes[-1]
  taps = w_shape[0] * w_shape[1]
  w_last = np.zeros(w_shape)
  w_last[:, :, :, 0] = (w_ci / taps)[None, None, :]
  b_last = np.zeros(b_shape)
  b_last[0] = b / n
  n_last = int(np.prod(w_shape)) + int(np.prod(b_shape))
  theta[theta.size - n_last:] = np.concatenate([w_last.ravel(), b_last.ravel()])
  check = vo.conv_forward(theta, cfgs, ansatz, geom, num_layers, nonlinearity, np.float64)
  assert np.abs(check - np.log(vec)).max() < 1e-8
  return theta.astype(np.float32), float(e0), cfgs.astype(np.float32), vec
