"""CPU restatement (numpy) of the cgs-vmc batched VMC inner loop.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference (/root/reference/cgs_vmc) is Python on TensorFlow 1.x +
Sonnet v1, neither of which can be imported here, and it ships no tests, golden vectors
or fixtures.  This file follows the reference op for op (citations below are
file:line under /root/reference/cgs_vmc) and is pinned instead by closed-form /
exact-diagonalisation / autograd / finite-difference checks in tests/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product path (cgs_vmc_amd) never does.

Parameter vector layout (a9, wavefunctions.py:167-175 creation order of snt.Linear
variables): [w_1 (N,H), b_1 (H), w_2 (H,H), b_2 (H), ..., w_out (H,1), b_out (1)],
each row-major, w:[in,out].

All functions take a `dtype` (np.float32 = the reference's arithmetic type;
np.float64 = the high-precision twin used to state tolerances).
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------- #
# Activations (layers.py:13-21)
# --------------------------------------------------------------------------- #
NONLINEARITIES = {
    'relu': lambda x: np.maximum(x, 0),
    'exp': np.exp,
    'cos': np.cos,
    'tan': np.tan,
    'tanh': np.tanh,
    'sigmoid': lambda x: 1.0 / (1.0 + np.exp(-x)),
    'identity': lambda x: x,
}

_NONLIN_DERIV = {
    'relu': lambda z, a: (z > 0).astype(z.dtype),
    'tanh': lambda z, a: 1 - a * a,
    'sigmoid': lambda z, a: a * (1 - a),
    'identity': lambda z, a: np.ones_like(z),
    'exp': lambda z, a: a,
    'cos': lambda z, a: -np.sin(z),
    'tan': lambda z, a: 1 + a * a,
}


# --------------------------------------------------------------------------- #
# Parameters
# --------------------------------------------------------------------------- #
def param_shapes(n_sites, layer_size, num_layers):
  """Shapes of FullyConnectedNetwork variables in creation order.

  wavefunctions.py:345-349: num_layers x [Linear(layer_size), act], Linear(1).
  """
  shapes = []
  fan_in = n_sites
  for _ in range(num_layers):
    shapes += [(fan_in, layer_size), (layer_size,)]
    fan_in = layer_size
  shapes += [(fan_in, 1), (1,)]
  return shapes


def num_params(n_sites, layer_size, num_layers):
  return int(sum(int(np.prod(s)) for s in param_shapes(n_sites, layer_size, num_layers)))


def unpack(theta, n_sites, layer_size, num_layers):
  """Splits a flat parameter vector into [(w, b), ...] (views)."""
  out, off = [], 0
  shapes = param_shapes(n_sites, layer_size, num_layers)
  for k in range(0, len(shapes), 2):
    ws, bs = shapes[k], shapes[k + 1]
    nw, nb = int(np.prod(ws)), int(np.prod(bs))
    w = theta[off:off + nw].reshape(ws); off += nw
    b = theta[off:off + nb].reshape(bs); off += nb
    out.append((w, b))
  assert off == theta.size
  return out


def init_params(n_sites, layer_size, num_layers, rng):
  """Sonnet-v1 snt.Linear default init restated: w ~ truncated normal (|x|<2 sigma)
  with sigma = 1/sqrt(fan_in), b = 0 (third-party semantics)."""
  parts = []
  for shp in param_shapes(n_sites, layer_size, num_layers):
    if len(shp) == 2:
      sigma = 1.0 / np.sqrt(shp[0])
      w = rng.standard_normal(shp)
      bad = np.abs(w) > 2
      while bad.any():
        w[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(w) > 2
      parts.append((w * sigma).ravel())
    else:
      parts.append(np.zeros(shp).ravel())
  return np.concatenate(parts).astype(np.float32)


# --------------------------------------------------------------------------- #
# Ansatz: RestrictedBoltzmannNetwork (wavefunctions.py:391-452)
# --------------------------------------------------------------------------- #
def rbm_param_shapes(n_sites, layer_size, num_layers):
  """Variables in creation order.  Sonnet v1 creates a Linear's variables when the module is
  first connected; _build connects the onsite layer first (wavefunctions.py:436), then the
  Sequential of num_layers x [Linear(H), act] + Linear(H) (wavefunctions.py:414-419)."""
  shapes = [(n_sites, 1), (1,)]
  fan_in = n_sites
  for _ in range(num_layers + 1):
    shapes += [(fan_in, layer_size), (layer_size,)]
    fan_in = layer_size
  return shapes


def rbm_num_params(n_sites, layer_size, num_layers):
  return int(sum(int(np.prod(s)) for s in rbm_param_shapes(n_sites, layer_size, num_layers)))


def rbm_unpack(theta, n_sites, layer_size, num_layers):
  out, off = [], 0
  shapes = rbm_param_shapes(n_sites, layer_size, num_layers)
  for k in range(0, len(shapes), 2):
    ws, bs = shapes[k], shapes[k + 1]
    nw, nb = int(np.prod(ws)), int(np.prod(bs))
    w = theta[off:off + nw].reshape(ws); off += nw
    b = theta[off:off + nb].reshape(bs); off += nb
    out.append((w, b))
  assert off == theta.size
  return out      # [onsite, layer_1, ..., layer_{L+1}]


def rbm_init_params(n_sites, layer_size, num_layers, rng):
  """snt.Linear default init (see init_params)."""
  parts = []
  for shp in rbm_param_shapes(n_sites, layer_size, num_layers):
    if len(shp) == 2:
      w = rng.standard_normal(shp)
      bad = np.abs(w) > 2
      while bad.any():
        w[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(w) > 2
      parts.append((w / np.sqrt(shp[0])).ravel())
    else:
      parts.append(np.zeros(shp).ravel())
  return np.concatenate(parts).astype(np.float32)


def rbm_logit(theta, configs, layer_size, num_layers, nonlinearity='relu', dtype=np.float32,
              return_acts=False):
  """onsite + sum_h log(cosh(z_h)) before the exp-normalisation shift
  (wavefunctions.py:414-420, 436-438).  log cosh is evaluated in its overflow-free form
  |z| + log1p(exp(-2|z|)) - log 2, which equals tf.log(tf.cosh(z)) wherever that is finite."""
  x = np.asarray(configs, dtype=dtype)
  layers = rbm_unpack(np.asarray(theta, dtype=dtype), x.shape[1], layer_size, num_layers)
  act = NONLINEARITIES[nonlinearity]
  (w_on, b_on), hidden = layers[0], layers[1:]
  onsite = (x @ w_on + b_on)[:, 0]                         # tf.squeeze(Linear(1)(inputs))
  zs, acts = [], [x]
  a = x
  for (w, b) in hidden[:-1]:
    z = a @ w + b
    a = act(z)
    zs.append(z); acts.append(a)
  w, b = hidden[-1]
  z_last = a @ w + b
  az = np.abs(z_last)
  logcosh = az + np.log1p(np.exp(-2 * az)) - dtype(np.log(2.0))
  logit = onsite + logcosh.sum(1, dtype=dtype)
  if return_acts:
    return logit, zs, acts, z_last
  return logit


def rbm_psi(theta, configs, layer_size, num_layers, shift=-10.0, nonlinearity='relu',
            output_activation='exp', dtype=np.float32):
  """psi = exp(onsite + (sum log cosh - shift)) (wavefunctions.py:419-420, 438)."""
  logit = rbm_logit(theta, configs, layer_size, num_layers, nonlinearity, dtype)
  with np.errstate(over='ignore'):
    return np.exp(logit - dtype(shift))


def rbm_weighted_logit_grads(theta, configs, weights, layer_size, num_layers,
                             nonlinearity='relu', dtype=np.float32):
  """sum_b weights[b, c] * d logit_b / d theta -> [C, P] in rbm_param_shapes order."""
  x = np.asarray(configs, dtype=dtype)
  w_b = np.asarray(weights, dtype=dtype)
  if w_b.ndim == 1:
    w_b = w_b[:, None]
  th = np.asarray(theta, dtype=dtype)
  layers = rbm_unpack(th, x.shape[1], layer_size, num_layers)
  hidden = layers[1:]
  _, zs, acts, z_last = rbm_logit(th, x, layer_size, num_layers, nonlinearity, dtype, True)
  dact = _NONLIN_DERIV[nonlinearity]
  out = []
  for c in range(w_b.shape[1]):
    wc = w_b[:, c:c + 1]
    grads = []
    delta = np.tanh(z_last)                                # d sum log cosh / d z_last
    for l in range(num_layers, -1, -1):
      grads = [acts[l].T @ (delta * wc), (delta * wc).sum(0)] + grads
      if l > 0:
        delta = (delta @ hidden[l][0].T) * dact(zs[l - 1], acts[l])
    grads = [x.T @ wc, wc.sum(0)] + grads                  # onsite w, b
    out.append(np.concatenate([g.ravel() for g in grads]))
  return np.stack(out)




# --------------------------------------------------------------------------- #
# Ansatz: Conv2DNetwork (wavefunctions.py:531-615) and ResNet2D (wavefunctions.py:710-809) on
# layers.Conv2dPeriodic / layers.ResBlock2d (layers.py:89-229).
#
# Geometry argument `geom` = (num_filters, kernel_size, size_x, size_y); it travels in the
# `layer_size` slot of the ANSATZ call signature.  `num_layers` is hparams.num_conv_layers
# (conv_2d) or hparams.num_resnet_blocks (res_net_2d).
#
# Third-party semantics restated (Sonnet v1 snt.Conv2D, TF1 tf.nn.conv2d / tf.nn.selu; sources
# not under /root/reference): w has shape [k, k, in_channels, out_channels], b [out_channels];
# tf.nn.conv2d is a cross-correlation; default initialisers are truncated normal with
# sigma = 1/sqrt(k*k*in_channels) for w and zeros for b; variables are created w then b when the
# module is first connected; selu(x) = 1.0507009873554805 * (x if x > 0 else
# 1.6732632423543772 * (exp(x) - 1)).
# --------------------------------------------------------------------------- #
SELU_SCALE = 1.0507009873554804934193349852946
SELU_ALPHA = 1.6732632423543772848170429916717


def selu(x):
  with np.errstate(over='ignore'):
    return x.dtype.type(SELU_SCALE) * np.where(
        x > 0, x, x.dtype.type(SELU_ALPHA) * (np.exp(np.minimum(x, 0)) - 1))


def selu_deriv(x):
  return x.dtype.type(SELU_SCALE) * np.where(
      x > 0, x.dtype.type(1), x.dtype.type(SELU_ALPHA) * np.exp(np.minimum(x, 0)))


# The 1-D types -- Conv1DNetwork (wavefunctions.py:455-527) and ResNet1D (wavefunctions.py:618-707)
# on layers.Conv1dPeriodic / ResBlock1d (layers.py:24-86, 229-293) -- are the same networks on a
# [N, 1] "lattice" with k x 1 kernels (snt.Conv1D variables w[k, in, out], b[out]) and the 1-D
# padding rule, which for an even kernel is the mirror image of the 2-D one (layers.py:66-72:
# k/2 in front, k/2 - 1 behind).  geom = (filters, kernel, N, 1).
CONV_1D = ('conv_1d', 'res_net_1d')
CONV_PLAIN = ('conv_1d', 'conv_2d')


def conv_layer_channels(ansatz, num_layers, num_filters):
  """(in_channels, out_channels) of every periodic convolution in creation order.
  conv_*: num_layers convolutions (wavefunctions.py:572-575, 487-490); res_net_*: the initial
  convolution, then first_conv / second_conv of each block (wavefunctions.py:766-772, 659-667;
  layers.py:200-201, 265-266)."""
  n_conv = num_layers if ansatz in CONV_PLAIN else 1 + 2 * num_layers
  return [(1 if l == 0 else num_filters, num_filters) for l in range(n_conv)]


def conv_param_shapes(ansatz, geom, num_layers):
  """Kernels as [k1, k2, in, out] (k2 = 1 for the 1-D types: the flat order equals snt.Conv1D's
  [k, in, out])."""
  f, k = geom[0], geom[1]
  k2 = 1 if ansatz in CONV_1D else k
  shapes = []
  for cin, cout in conv_layer_channels(ansatz, num_layers, f):
    shapes += [(k, k2, cin, cout), (cout,)]
  return shapes


def conv_num_params(ansatz, geom, num_layers):
  return int(sum(int(np.prod(s)) for s in conv_param_shapes(ansatz, geom, num_layers)))


def conv_unpack(theta, ansatz, geom, num_layers):
  out, off = [], 0
  shapes = conv_param_shapes(ansatz, geom, num_layers)
  for i in range(0, len(shapes), 2):
    nw, nb = int(np.prod(shapes[i])), int(np.prod(shapes[i + 1]))
    w = theta[off:off + nw].reshape(shapes[i]); off += nw
    b = theta[off:off + nb]; off += nb
    out.append((w, b))
  assert off == theta.size
  return out


def conv_init_params(ansatz, geom, num_layers, rng):
  """snt.Conv2D default init: truncated normal, sigma = 1/sqrt(k*k*in_channels); b = 0."""
  parts = []
  for shp in conv_param_shapes(ansatz, geom, num_layers):
    if len(shp) == 4:
      w = rng.standard_normal(shp)
      bad = np.abs(w) > 2
      while bad.any():
        w[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(w) > 2
      parts.append((w / np.sqrt(shp[0] * shp[1] * shp[2])).ravel())
    else:
      parts.append(np.zeros(shp).ravel())
  return np.concatenate(parts).astype(np.float32)


def periodic_pad_2d(x, k):
  """layers.Conv2dPeriodic._pad_input (layers.py:118-148) on x [B, D1, D2, C]: axis 2 gets
  (k-1)//2 columns of the far end in front and k//2 of the near end behind, then axis 1 the
  same (odd k: (k-1)/2 both; even k: k/2 - 1 in front and k/2 behind)."""
  lo = (k - 1) // 2 if k % 2 == 1 else k // 2 - 1
  hi = (k - 1) // 2 if k % 2 == 1 else k // 2
  d1, d2 = x.shape[1], x.shape[2]
  left = x[:, :, d2 - lo:]
  right = x[:, :, :hi]
  wp = np.concatenate([left, x, right], axis=2)
  bot = wp[:, d1 - lo:]
  top = wp[:, :hi]
  return np.concatenate([bot, wp, top], axis=1)


def periodic_pad_1d(x, k):
  """layers.Conv1dPeriodic._pad_input (layers.py:51-74) on x [B, N, 1, C] (axis 1): (k-1)/2 on both
  sides for odd k; for even k, k/2 in front and k/2 - 1 behind."""
  lo = (k - 1) // 2 if k % 2 == 1 else k // 2
  hi = (k - 1) // 2 if k % 2 == 1 else k // 2 - 1
  n = x.shape[1]
  return np.concatenate([x[:, n - lo:], x, x[:, :hi]], axis=1)


def conv2d_periodic(x, w, b):
  """Conv2dPeriodic._build (layers.py:151-160) / Conv1dPeriodic._build (layers.py:77-86):
  snt.Conv2D / Conv1D (VALID, stride 1) of the padded input = cross-correlation
  out[a1,a2,o] = b[o] + sum w[d1,d2,c,o] pad[a1+d1, a2+d2, c].  A kernel of shape [k, 1, ., .]
  is the 1-D module acting along axis 1."""
  k1, k2 = w.shape[0], w.shape[1]
  pad = periodic_pad_2d(x, k1) if k2 == k1 and x.shape[2] > 1 or k2 > 1 else periodic_pad_1d(x, k1)
  bsz, d1, d2 = x.shape[0], x.shape[1], x.shape[2]
  out = np.zeros((bsz, d1, d2, w.shape[3]), x.dtype)
  for i in range(k1):
    for j in range(k2):
      out += pad[:, i:i + d1, j:j + d2, :] @ w[i, j]
  return out + b


def conv2d_periodic_backward(x, w, delta):
  """(d/dx, d/dw, d/db) of sum(conv2d_periodic(x, w, b) * delta)."""
  k1, k2 = w.shape[0], w.shape[1]
  one_d = k2 == 1 and x.shape[2] == 1
  lo1 = k1 // 2 if one_d else (k1 - 1) // 2       # padding in front (even 1-D kernels: k/2)
  lo2 = 0 if one_d else (k2 - 1) // 2
  dx = np.zeros_like(x)
  dw = np.zeros_like(w)
  for i in range(k1):
    for j in range(k2):
      # pad[a1+i, a2+j] = x[(a1+i-lo1) mod D1, (a2+j-lo2) mod D2]
      xs = np.roll(x, (-(i - lo1), -(j - lo2)), axis=(1, 2))
      dw[i, j] = np.tensordot(xs, delta, axes=([0, 1, 2], [0, 1, 2]))
      dx += np.roll(delta @ w[i, j].T, (i - lo1, j - lo2), axis=(1, 2))
  return dx, dw, delta.sum((0, 1, 2))


def conv_forward(theta, configs, ansatz, geom, num_layers, nonlinearity='relu',
                 dtype=np.float32, return_tape=False):
  """Pre-output-activation scalar of Conv2DNetwork / ResNet2D: reduce_sum over sites and
  channels of the last feature map (wavefunctions.py:569, 577; 760, 773)."""
  f, k, sx, sy = geom
  x = np.asarray(configs, dtype=dtype).reshape(-1, sx, sy, 1)   # wavefunctions.py:596-597 (1-D: expand_dims, :511)
  layers_ = conv_unpack(np.asarray(theta, dtype=dtype), ansatz, geom, num_layers)
  tape = []   # per convolution: (input, pre-activation output)
  if ansatz in CONV_PLAIN:
    act = NONLINEARITIES[nonlinearity]
    a = x
    for l, (w, b) in enumerate(layers_):
      z = conv2d_periodic(a, w, b)
      tape.append((a, z))
      a = act(z) if l + 1 != len(layers_) else z            # wavefunctions.py:574-575
    last = a
  elif ansatz in ('res_net_2d', 'res_net_1d'):
    w, b = layers_[0]
    h = conv2d_periodic(x, w, b)                             # initial_conv, no activation
    tape.append((x, h))
    for blk in range(num_layers):                            # layers.py:226-228
      (w1, b1), (w2, b2) = layers_[1 + 2 * blk], layers_[2 + 2 * blk]
      u = conv2d_periodic(h, w1, b1)
      t = selu(u)
      v = conv2d_periodic(t, w2, b2)
      tape.append((h, u)); tape.append((t, v))
      h = v + h
    last = h
  else:
    raise ValueError(ansatz)
  logit = last.reshape(last.shape[0], -1).sum(1, dtype=dtype)
  if return_tape == 'scale':       # sum |entries|: the rounding scale of an fp32 evaluation of the sum
    return logit, np.abs(last).reshape(last.shape[0], -1).sum(1)
  if return_tape:
    return logit, tape, layers_
  return logit


def _conv_psi(ansatz):
  def psi(theta, configs, geom, num_layers, shift=-10.0, nonlinearity='relu',
          output_activation='exp', dtype=np.float32):
    logit = conv_forward(theta, configs, ansatz, geom, num_layers, nonlinearity, dtype)
    if output_activation == 'exp':                           # wavefunctions.py:576-579
      with np.errstate(over='ignore'):
        return np.exp(logit - dtype(shift))
    return NONLINEARITIES[output_activation](logit)
  return psi


def _conv_logit(ansatz):
  def logit(theta, configs, geom, num_layers, nonlinearity='relu', dtype=np.float32):
    return conv_forward(theta, configs, ansatz, geom, num_layers, nonlinearity, dtype)
  return logit


def _conv_weighted_grads(ansatz):
  def grads(theta, configs, weights, geom, num_layers, nonlinearity='relu', dtype=np.float32,
            output_activation='exp'):
    """sum_b weights[b, c] * d logit_b / d theta -> [C, P] (manual back-propagation)."""
    w_b = np.asarray(weights, dtype=dtype)
    if w_b.ndim == 1:
      w_b = w_b[:, None]
    logit, tape, layers_ = conv_forward(theta, configs, ansatz, geom, num_layers, nonlinearity,
                                        dtype, True)
    w_b = w_b * output_dlog(logit, output_activation, dtype)[:, None]
    out = []
    for c in range(w_b.shape[1]):
      wc = w_b[:, c][:, None, None, None]
      grads_ = [None] * len(layers_)
      if ansatz in CONV_PLAIN:
        dact = _NONLIN_DERIV[nonlinearity]
        delta = np.broadcast_to(wc, tape[-1][1].shape).astype(dtype)   # d logit / d z_last = 1
        for l in range(len(layers_) - 1, -1, -1):
          a_in, _ = tape[l]
          dx, dw, db = conv2d_periodic_backward(a_in, layers_[l][0], delta)
          grads_[l] = (dw, db)
          if l > 0:
            z_prev = tape[l - 1][1]
            delta = dx * dact(z_prev, NONLINEARITIES[nonlinearity](z_prev))
      else:
        dh = np.broadcast_to(wc, tape[-1][1].shape).astype(dtype)
        for blk in range(num_layers - 1, -1, -1):
          (h_in, u), (t, _) = tape[1 + 2 * blk], tape[2 + 2 * blk]
          dt, dw2, db2 = conv2d_periodic_backward(t, layers_[2 + 2 * blk][0], dh)
          du = dt * selu_deriv(u)
          dhin, dw1, db1 = conv2d_periodic_backward(h_in, layers_[1 + 2 * blk][0], du)
          grads_[2 + 2 * blk] = (dw2, db2); grads_[1 + 2 * blk] = (dw1, db1)
          dh = dh + dhin
        _, dw0, db0 = conv2d_periodic_backward(tape[0][0], layers_[0][0], dh)
        grads_[0] = (dw0, db0)
      out.append(np.concatenate([np.concatenate([g[0].ravel(), g[1].ravel()]) for g in grads_]))
    return np.stack(out)
  return grads


# --------------------------------------------------------------------------- #
# Ansatz: FullyConnectedNetwork (wavefunctions.py:328-371) + exp shift (206-232)
# --------------------------------------------------------------------------- #
def fc_logit(theta, configs, layer_size, num_layers, nonlinearity='relu',
             dtype=np.float32, return_acts=False):
  """Pre-exp output of the network: squeeze(Linear(1)(act(Linear(...)))).

  wavefunctions.py:345-349, 370-371.  `configs` [B,N] of +-1.
  """
  x = np.asarray(configs, dtype=dtype)
  layers = unpack(np.asarray(theta, dtype=dtype), x.shape[1], layer_size, num_layers)
  act = NONLINEARITIES[nonlinearity]
  zs, acts = [], [x]
  a = x
  for (w, b) in layers[:-1]:
    z = a @ w + b
    a = act(z)
    zs.append(z); acts.append(a)
  w, b = layers[-1]
  logit = (a @ w + b)[:, 0]
  if return_acts:
    return logit, zs, acts
  return logit


def fc_psi(theta, configs, layer_size, num_layers, shift=-10.0, nonlinearity='relu',
           output_activation='exp', dtype=np.float32):
  """psi = exp(logit - exp_norm_shift) (wavefunctions.py:350-353, 232); shift starts
  at -10 (wavefunctions.py:209)."""
  logit = fc_logit(theta, configs, layer_size, num_layers, nonlinearity, dtype)
  if output_activation == 'exp':
    with np.errstate(over='ignore'):
      return np.exp(logit - dtype(shift))
  return NONLINEARITIES[output_activation](logit)


def update_norm(psi, shift, max_value=1e10):
  """wavefunctions.py:261-288: if log(max psi) > log(max_value) the shift grows by the
  excess, otherwise += 0."""
  max_log = np.log(max_value)
  log_max = np.log(np.max(psi))
  if log_max > max_log:
    return np.float32(shift + (log_max - max_log))
  return np.float32(shift)


def normalize_batch(psi, shift, max_value=1e10):
  """wavefunctions.py:234-259."""
  return np.float32(shift + (np.log(np.max(psi)) - np.log(max_value)))


# --------------------------------------------------------------------------- #
# Initial chains (utils.py:169-192)
# --------------------------------------------------------------------------- #
def random_configurations(n_sites, batch_size, rng):
  """All +1, then n_sites//2 distinct random sites set to -1 by rejection; float32.
  The reference is unseeded (np.random.RandomState()); here the caller owns `rng`
  (np.random.RandomState-compatible .randint)."""
  configurations = np.ones((batch_size, n_sites))
  for i in range(batch_size):
    pos = rng.randint(0, n_sites)
    for _ in range(n_sites // 2):
      while configurations[i, pos] != 1.0:
        pos = rng.randint(0, n_sites)
      configurations[i, pos] = -1.0
  return configurations.astype(np.float32)


# --------------------------------------------------------------------------- #
# Counter-based RNG: Philox4x32-10 (Salmon et al., SC'11; Random123 v1.14 KAT pinned
# in tests/test_oracle_rng.py).  The reference uses tf.random_uniform (unseeded), so
# the generator is this build's own choice; the *use* of the uniforms follows
# graph_builders.py:59-65,76-77.
# --------------------------------------------------------------------------- #
_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = np.uint32(0x9E3779B9)
_PHILOX_W1 = np.uint32(0xBB67AE85)
ACCEPT_BLOCK = 0xFFFFFFFF  # counter word 0 of the acceptance draw


def philox4x32_10(c0, c1, c2, c3, k0, k1):
  """Vectorised Philox4x32-10.  Inputs broadcastable uint32 arrays."""
  c0, c1, c2, c3, k0, k1 = np.broadcast_arrays(
      *[np.asarray(v, dtype=np.uint32) for v in (c0, c1, c2, c3, k0, k1)])
  c0, c1, c2, c3, k0, k1 = [v.copy() for v in (c0, c1, c2, c3, k0, k1)]
  mask = np.uint64(0xFFFFFFFF)
  with np.errstate(over='ignore'):
    for r in range(10):
      p0 = _PHILOX_M0 * c0.astype(np.uint64)
      p1 = _PHILOX_M1 * c2.astype(np.uint64)
      hi0 = (p0 >> np.uint64(32)).astype(np.uint32); lo0 = (p0 & mask).astype(np.uint32)
      hi1 = (p1 >> np.uint64(32)).astype(np.uint32); lo1 = (p1 & mask).astype(np.uint32)
      c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
      if r < 9:
        k0 = (k0 + _PHILOX_W0).astype(np.uint32)
        k1 = (k1 + _PHILOX_W1).astype(np.uint32)
  return c0, c1, c2, c3


def u32_to_uniform(x):
  """[0,1) float32 with 24 random bits (tf.random_uniform is [0,1) fp32)."""
  return ((np.asarray(x, dtype=np.uint32) >> np.uint32(8)).astype(np.float32)
          * np.float32(1.0 / 16777216.0))


def step_uniforms(seed, chain_ids, step, n_sites):
  """Uniforms of one mc_step: U[B,N] (site draws, graph_builders.py:59) and u[B]
  (acceptance draw, graph_builders.py:76-77).

  Counter = (block, global chain id, step_lo, step_hi), key = (seed_lo, seed_hi);
  block b yields sites 4b..4b+3, block 0xFFFFFFFF word 0 the acceptance draw.
  """
  chain_ids = np.asarray(chain_ids, dtype=np.uint32)
  nblk = (n_sites + 3) // 4
  blocks = np.arange(nblk, dtype=np.uint32)[None, :]
  k0 = np.uint32(seed & 0xFFFFFFFF); k1 = np.uint32((seed >> 32) & 0xFFFFFFFF)
  s_lo = np.uint32(step & 0xFFFFFFFF); s_hi = np.uint32((step >> 32) & 0xFFFFFFFF)
  r = philox4x32_10(blocks, chain_ids[:, None], s_lo, s_hi, k0, k1)
  sites = np.stack(r, axis=-1).reshape(len(chain_ids), nblk * 4)[:, :n_sites]
  ra = philox4x32_10(np.uint32(ACCEPT_BLOCK), chain_ids, s_lo, s_hi, k0, k1)
  return u32_to_uniform(sites), u32_to_uniform(ra[0])


# --------------------------------------------------------------------------- #
# Metropolis exchange step (graph_builders.py:38-89)
# --------------------------------------------------------------------------- #
def propose_exchange(configs, site_uniforms):
  """graph_builders.py:59-65: swap_choice = configs * u; the DOWN spin to raise is
  argmin (most negative = down spin with the largest u), the UP spin to lower is
  argmax.  np.argmin/argmax return the first index on ties like tf.argmin/argmax."""
  swap_choice = np.asarray(configs, np.float32) * np.asarray(site_uniforms, np.float32)
  return np.argmax(swap_choice, axis=1), np.argmin(swap_choice, axis=1)  # (i_up, i_dn)


def mc_step(amp_fn, configs, i_up, i_dn, u_accept):
  """One exchange proposal + Metropolis accept per chain, reference structure
  (two forward passes, graph_builders.py:54-55 and 74).

  amp_fn(configs)->psi.  Returns (new_configs, accept_mask, ratios).
  accept = |psi'|/|psi| > sqrt(u)   (graph_builders.py:75-79, strict >).
  """
  configs = np.asarray(configs, np.float32)
  rows = np.arange(configs.shape[0])
  psi = amp_fn(configs)
  updated = configs.copy()
  np.add.at(updated, (rows, i_dn), np.float32(2.0))    # graph_builders.py:67-68
  np.add.at(updated, (rows, i_up), np.float32(-2.0))   # graph_builders.py:70-71
  new_psi = amp_fn(updated)
  with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
    ratios = np.abs(new_psi) / np.abs(psi)
  rnd = np.sqrt(np.asarray(u_accept, np.float32))
  accept = ratios > rnd
  out = configs.copy()
  out[accept] = updated[accept]
  return out, accept, ratios


# --------------------------------------------------------------------------- #
# Heisenberg operator (operators.py:128-287)
# --------------------------------------------------------------------------- #
def heisenberg_build(amp_fn, configs, bonds, j_x, j_z, dtype=np.float32):
  """HeisenbergHamiltonian.build (operators.py:227-247): sum over bonds of
  HeisenbergBond.build (operators.py:137-169).

  Per bond: diag = 0.25*jz*s_i*s_j; off = 0.25*jx*2*[s_i*s_j<0]*psi(R with s_i,s_j
  swapped).  The swapped forward runs on ALL rows and is masked afterwards.
  j_x / j_z may be scalars (the reference) or per-bond arrays (extension D5).
  """
  x = np.asarray(configs, dtype=dtype)
  nb = len(bonds)
  jx = np.broadcast_to(np.asarray(j_x, dtype=dtype), (nb,))
  jz = np.broadcast_to(np.asarray(j_z, dtype=dtype), (nb,))
  diag = np.zeros(x.shape[0], dtype=dtype)
  off = np.zeros(x.shape[0], dtype=dtype)
  for k, (i, j) in enumerate(bonds):
    si, sj = x[:, i], x[:, j]
    upd = x.copy()
    upd[:, i] += sj - si            # operators.py:162
    upd[:, j] += si - sj            # operators.py:163
    sz = si * sj                    # operators.py:165
    mask = (sz < 0).astype(dtype)   # operators.py:166-167
    perp = dtype(2.0) * mask * amp_fn(upd).astype(dtype)   # operators.py:168
    diag += dtype(0.25) * jz[k] * sz
    off += dtype(0.25) * jx[k] * perp
  return diag, off


def local_value(amp_fn, configs, bonds, j_x, j_z, psi=None, dtype=np.float32):
  """operators.py:249-259: diag + off / psi."""
  if psi is None:
    psi = amp_fn(configs)
  diag, off = heisenberg_build(amp_fn, configs, bonds, j_x, j_z, dtype)
  return diag + off / np.asarray(psi, dtype)


def apply_in_place(amp_fn, configs, bonds, j_x, j_z, psi=None, dtype=np.float32):
  """operators.py:261-271: diag * psi + off."""
  if psi is None:
    psi = amp_fn(configs)
  diag, off = heisenberg_build(amp_fn, configs, bonds, j_x, j_z, dtype)
  return diag * np.asarray(psi, dtype) + off


def constant_psi_local_energy(configs, bonds, j_x, j_z):
  """Closed form for psi == const: 0.25*jz*(n_par - n_anti) + 0.5*jx*n_anti."""
  x = np.asarray(configs, np.float64)
  jx = np.broadcast_to(np.asarray(j_x, np.float64), (len(bonds),))
  jz = np.broadcast_to(np.asarray(j_z, np.float64), (len(bonds),))
  e = np.zeros(x.shape[0])
  for k, (i, j) in enumerate(bonds):
    sz = x[:, i] * x[:, j]
    e += 0.25 * jz[k] * sz + 0.5 * jx[k] * (sz < 0)
  return e


# --------------------------------------------------------------------------- #
# d logit / d theta, batch-summed with weights (manual back-prop)
# --------------------------------------------------------------------------- #
def output_dlog(logit, output_activation, dtype):
  """(1/psi) d psi / d x for psi = g(x): the factor tf.gradients(psi / stop_gradient(psi))
  (training.py:545) puts in front of d x / d theta.  1 for the exp output."""
  if output_activation == 'exp':
    return np.ones_like(logit)
  g = NONLINEARITIES[output_activation](logit)
  with np.errstate(divide='ignore', invalid='ignore'):
    return (_NONLIN_DERIV[output_activation](logit, g) / g).astype(dtype)


def weighted_logit_grads(theta, configs, weights, layer_size, num_layers,
                         nonlinearity='relu', dtype=np.float32, output_activation='exp'):
  """Returns sum_b weights[b, c] * d logit_b / d theta for every column c of
  `weights` [B, C] -> [C, P].

  This is tf.gradients(psi/stop_gradient(psi) * w, opt_v) of training.py:545-547 and
  674-679: d(psi_b/psi_ng_b)/d theta = d logit_b / d theta for an exp output, and
  tf.gradients sums over the batch.
  """
  x = np.asarray(configs, dtype=dtype)
  w_b = np.asarray(weights, dtype=dtype)
  if w_b.ndim == 1:
    w_b = w_b[:, None]
  th = np.asarray(theta, dtype=dtype)
  layers = unpack(th, x.shape[1], layer_size, num_layers)
  logit, zs, acts = fc_logit(th, x, layer_size, num_layers, nonlinearity, dtype, True)
  # a non-exp output activation g: d (psi/psi_ng) = (g'(x)/g(x)) d x, a per-sample factor
  w_b = w_b * output_dlog(logit, output_activation, dtype)[:, None]
  dact = _NONLIN_DERIV[nonlinearity]
  n_cols = w_b.shape[1]
  grads = [[] for _ in range(n_cols)]
  # output layer
  w_out, _ = layers[-1]
  a_last = acts[-1]
  for c in range(n_cols):
    grads[c] = [(a_last * w_b[:, c:c + 1]).sum(0)[:, None], w_b[:, c].sum(keepdims=True)]
  delta = np.broadcast_to(w_out[:, 0][None, :], a_last.shape).copy()  # d logit / d a_L
  for l in range(num_layers - 1, -1, -1):
    delta = delta * dact(zs[l], acts[l + 1])          # d logit / d z_l   [B,H]
    a_prev = acts[l]
    for c in range(n_cols):
      dw = a_prev.T @ (delta * w_b[:, c:c + 1])
      db = (delta * w_b[:, c:c + 1]).sum(0)
      grads[c] = [dw, db] + grads[c]
    if l > 0:
      delta = delta @ layers[l][0].T
  return np.stack([np.concatenate([g.ravel() for g in grads[c]]) for c in range(n_cols)])


# --------------------------------------------------------------------------- #
# EnergyGradient accumulators + gradient (training.py:539-567)
# --------------------------------------------------------------------------- #
class Accumulators:
  """tf.metrics.mean / mean_tensor state (local variables).

  mean(values): total += sum(values), count += size.
  mean_tensor(values): total += values (elementwise), count += 1.
  """

  def __init__(self, n_params, dtype=np.float32):
    self.dtype = dtype
    self.n_params = n_params
    self.reset()

  def reset(self):
    """training.py:568 / 707: tf.variables_initializer(tf.local_variables())."""
    z = lambda: np.zeros(self.n_params, self.dtype)
    self.g1_total, self.g2_total = z(), z()
    self.g_count = self.dtype(0)
    self.e_total = self.dtype(0); self.e_count = self.dtype(0)
    self.r_total = self.dtype(0); self.r_count = self.dtype(0)

  def mean_energy(self):
    return self.e_total / self.e_count

  def mean_ratio(self):
    return self.r_total / self.r_count


def _act_kwargs(ansatz, nonlinearity, output_activation):
  kw = {'nonlinearity': nonlinearity}
  if ansatz in ('fully_connected', 'conv_2d', 'res_net_2d', 'conv_1d', 'res_net_1d'):
    kw['output_activation'] = output_activation
  return kw


def energy_gradient_accumulate(acc, theta, configs, bonds, j_x, j_z, shift,
                               layer_size, num_layers, dtype=np.float32,
                               ansatz='fully_connected', nonlinearity='relu',
                               output_activation='exp'):
  """One `accumulate_gradients` run of EnergyGradientOptimizer (training.py:539-558)."""
  psi_fn = ANSATZ[ansatz][0]
  grads_fn = ANSATZ[ansatz][2] or weighted_logit_grads
  kw = _act_kwargs(ansatz, nonlinearity, output_activation)
  amp = lambda c: psi_fn(theta, c, layer_size, num_layers, shift, dtype=dtype, **kw)
  psi = amp(configs)
  e_loc = local_value(amp, configs, bonds, j_x, j_z, psi, dtype)     # 542-543
  ones = np.ones_like(e_loc)
  g = grads_fn(theta, configs, np.stack([ones, e_loc], 1),
               layer_size, num_layers, dtype=dtype, **kw)            # 545-547
  acc.g1_total += g[0]; acc.g2_total += g[1]; acc.g_count += 1        # 550-553
  acc.e_total += e_loc.sum(dtype=dtype); acc.e_count += e_loc.size    # 555
  return e_loc


def energy_gradient(acc):
  """training.py:560-564: mean(scaled) - mean(E) * mean(pure)."""
  return acc.g2_total / acc.g_count - acc.mean_energy() * (acc.g1_total / acc.g_count)


# --------------------------------------------------------------------------- #
# Stochastic reconfiguration (EXTENSION named by the north star; NOT in the reference:
# training.py only has the plain gradient + Adam).  fp64 explicit-S restatement of what
# cgs_vmc_amd/csrc/sr.hip solves matrix-free; it is the only oracle this path has.
# --------------------------------------------------------------------------- #
def per_sample_logit_grads(theta, configs, layer_size, num_layers, nonlinearity='relu',
                           dtype=np.float64):
  """O[b, k] = d logit_b / d theta_k  -> [B, P] (same back-prop as weighted_logit_grads,
  kept per sample)."""
  x = np.asarray(configs, dtype=dtype)
  th = np.asarray(theta, dtype=dtype)
  layers = unpack(th, x.shape[1], layer_size, num_layers)
  _, zs, acts = fc_logit(th, x, layer_size, num_layers, nonlinearity, dtype, True)
  dact = _NONLIN_DERIV[nonlinearity]
  b = x.shape[0]
  pieces = [acts[-1], np.ones((b, 1), dtype)]
  delta = np.broadcast_to(layers[-1][0][:, 0][None, :], acts[-1].shape).copy()
  for l in range(num_layers - 1, -1, -1):
    delta = delta * dact(zs[l], acts[l + 1])
    dw = (acts[l][:, :, None] * delta[:, None, :]).reshape(b, -1)
    pieces = [dw, delta.copy()] + pieces
    if l > 0:
      delta = delta @ layers[l][0].T
  return np.concatenate(pieces, axis=1)


def rbm_per_sample_logit_grads(theta, configs, layer_size, num_layers, nonlinearity='relu',
                               dtype=np.float64):
  """O[b, k] for the rbm ansatz (one-hot weights through rbm_weighted_logit_grads; small cases)."""
  b = np.asarray(configs).shape[0]
  return rbm_weighted_logit_grads(theta, configs, np.eye(b, dtype=dtype), layer_size, num_layers,
                                  nonlinearity, dtype)


def sr_system(o, e_loc):
  """S = <O O^T> - <O><O>^T and f = <E O> - <E><O> over the samples (rows of o)."""
  o = np.asarray(o, np.float64)
  e = np.asarray(e_loc, np.float64)
  n = o.shape[0]
  o_mean = o.mean(0)
  s = o.T @ o / n - np.outer(o_mean, o_mean)
  f = o.T @ e / n - e.mean() * o_mean
  return s, f


def sr_solve(o, e_loc, diag_shift):
  """x with (S + diag_shift I) x = f (dense solve)."""
  s, f = sr_system(o, e_loc)
  return np.linalg.solve(s + diag_shift * np.eye(s.shape[0]), f)


def sr_conjugate_gradient(o, e_loc, diag_shift, tol, max_iter):
  """The same CG recurrence as the HIP path (x0 = 0, stop at |r| <= tol |f|), with the
  matrix-free product S v = O^T (O v) / n - <O> mean(O v).  Returns (x, iterations)."""
  o = np.asarray(o, np.float64)
  e = np.asarray(e_loc, np.float64)
  n = o.shape[0]
  o_mean = o.mean(0)
  f = o.T @ e / n - e.mean() * o_mean
  x = np.zeros_like(f); r = f.copy(); p = f.copy()
  rr0 = rr = r @ r
  it = 0
  while it < max_iter and rr > tol * tol * rr0 and rr0 > 0:
    t = o @ p
    q = o.T @ t / n - o_mean * t.mean() + diag_shift * p
    alpha = rr / (p @ q)
    x += alpha * p
    r -= alpha * q
    rr_new = r @ r
    p = r + (rr_new / rr) * p
    rr = rr_new
    it += 1
  return x, it


# --------------------------------------------------------------------------- #
# LogOverlapImaginaryTimeSWO accumulators + gradient (training.py:652-699)
# --------------------------------------------------------------------------- #
def log_overlap_accumulate(acc, theta, theta_omega, configs, bonds, j_x, j_z, shift,
                           shift_omega, beta, layer_size, num_layers, dtype=np.float32,
                           ansatz='fully_connected', nonlinearity='relu',
                           output_activation='exp'):
  """One `accumulate_gradients` run of LogOverlapImaginaryTimeSWO.

  The supervisor omega is a deepcopy with its OWN exp_norm_shift variable, created at
  -10 and never updated (wavefunctions.py:177-204, 209; module_transfer_ops copies
  trainables only, 300-325).
  """
  psi_fn = ANSATZ[ansatz][0]
  grads_fn = ANSATZ[ansatz][2] or weighted_logit_grads
  kw = _act_kwargs(ansatz, nonlinearity, output_activation)
  amp = lambda c: psi_fn(theta, c, layer_size, num_layers, shift, dtype=dtype, **kw)
  amp_w = lambda c: psi_fn(theta_omega, c, layer_size, num_layers, shift_omega, dtype=dtype, **kw)
  psi = amp(configs)                                                   # 661
  psi_w = amp_w(configs)                                               # 662
  h_psi_w = apply_in_place(amp_w, configs, bonds, j_x, j_z, psi_w, dtype)  # 664
  ite = psi_w - dtype(beta) * h_psi_w                                  # 665-666
  e_loc = h_psi_w / psi_w                                              # 667
  ratio = ite / psi                                                    # 672
  ones = np.ones_like(ratio)
  g = grads_fn(theta, configs, np.stack([ones, ratio], 1),
               layer_size, num_layers, dtype=dtype, **kw)              # 674-679
  acc.g1_total += g[0]; acc.g2_total += g[1]; acc.g_count += 1
  acc.e_total += e_loc.sum(dtype=dtype); acc.e_count += e_loc.size     # 689
  acc.r_total += ratio.sum(dtype=dtype); acc.r_count += ratio.size     # 690
  return e_loc, ratio


def log_overlap_gradient(acc):
  """training.py:697-699: mean(log_grad) - mean(ratio_grad) / mean(ratio)."""
  return acc.g1_total / acc.g_count - (acc.g2_total / acc.g_count) / acc.mean_ratio()


# --------------------------------------------------------------------------- #
# Optimizer (training.py:76-91; TF1 AdamOptimizer semantics)
# --------------------------------------------------------------------------- #
def piecewise_constant(x, boundaries, values):
  """tf.train.piecewise_constant: values[0] if x <= b[0]; values[i] if
  b[i-1] < x <= b[i]; values[-1] if x > b[-1]."""
  for b, v in zip(boundaries, values):
    if x <= b:
      return v
  return values[-1]


class AdamState:
  def __init__(self, n_params):
    self.m = np.zeros(n_params, np.float32)
    self.v = np.zeros(n_params, np.float32)
    self.t = 0


def adam_apply(state, theta, grad, lr, beta1=0.9, beta2=0.99, eps=1e-8):
  """TF1 Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; theta -= lr_t*m/(sqrt(v)+eps)."""
  state.t += 1
  f = np.float32
  lr_t = f(lr) * np.sqrt(f(1) - f(beta2) ** f(state.t)) / (f(1) - f(beta1) ** f(state.t))
  g = np.asarray(grad, np.float32)
  state.m = state.m + (g - state.m) * f(1 - beta1)
  state.v = state.v + (g * g - state.v) * f(1 - beta2)
  return (np.asarray(theta, np.float32)
          - f(lr_t) * state.m / (np.sqrt(state.v) + f(eps))).astype(np.float32)


# --------------------------------------------------------------------------- #
# Lattices (the build's own helpers; run_training.py:103-109 gives the 1-D default)
# --------------------------------------------------------------------------- #
def chain_bonds(n_sites):
  """run_training.py:109 default: 1-D periodic chain."""
  return [(i, (i + 1) % n_sites) for i in range(n_sites)]


def torus_bonds(lx, ly, next_nearest=False):
  """L_x x L_y periodic square lattice, each bond once; site = x + lx*y."""
  bonds = []
  for y in range(ly):
    for x in range(lx):
      s = x + lx * y
      bonds.append((s, (x + 1) % lx + lx * y))
      bonds.append((s, x + lx * ((y + 1) % ly)))
  if next_nearest:
    for y in range(ly):
      for x in range(lx):
        s = x + lx * y
        bonds.append((s, (x + 1) % lx + lx * ((y + 1) % ly)))
        bonds.append((s, (x - 1) % lx + lx * ((y + 1) % ly)))
  return bonds


# --------------------------------------------------------------------------- #
# Loops with the reference's call structure (used as the timed CPU baseline)
# --------------------------------------------------------------------------- #
def run_sweeps(theta, configs, n_steps, seed, step0, layer_size, num_layers, shift=-10.0,
               chain_offset=0, dtype=np.float32, ansatz='fully_connected', nonlinearity='relu',
               output_activation='exp'):
  """n_steps mc_steps, one host-level call each with two forwards, as
  training.py:608-609 / evaluation.py:138-139 drive graph_builders.py:38-89."""
  kw = _act_kwargs(ansatz, nonlinearity, output_activation)
  amp = lambda c: ANSATZ[ansatz][0](theta, c, layer_size, num_layers, shift, dtype=dtype, **kw)
  ids = np.arange(configs.shape[0], dtype=np.uint32) + np.uint32(chain_offset)
  accepted = 0
  for t in range(n_steps):
    u_sites, u_acc = step_uniforms(seed, ids, step0 + t, configs.shape[1])
    i_up, i_dn = propose_exchange(configs, u_sites)
    configs, acc, _ = mc_step(amp, configs, i_up, i_dn, u_acc)
    accepted += int(acc.sum())
  return configs, accepted


# ansatz name -> (psi, logit, weighted grads, init, num_params); `ansatz=` of the functions below
ANSATZ = {
    'fully_connected': (fc_psi, fc_logit, None, init_params, num_params),
    'rbm': (rbm_psi, rbm_logit, rbm_weighted_logit_grads, rbm_init_params, rbm_num_params),
    # convolutional ansatz types: the `layer_size` slot carries geom = (filters, kernel, sx, sy)
    'conv_2d': (_conv_psi('conv_2d'), _conv_logit('conv_2d'), _conv_weighted_grads('conv_2d'),
                lambda geom, num_layers, rng: conv_init_params('conv_2d', geom, num_layers, rng),
                lambda geom, num_layers: conv_num_params('conv_2d', geom, num_layers)),
    'res_net_2d': (_conv_psi('res_net_2d'), _conv_logit('res_net_2d'),
                   _conv_weighted_grads('res_net_2d'),
                   lambda geom, num_layers, rng: conv_init_params('res_net_2d', geom, num_layers, rng),
                   lambda geom, num_layers: conv_num_params('res_net_2d', geom, num_layers)),
    'conv_1d': (_conv_psi('conv_1d'), _conv_logit('conv_1d'), _conv_weighted_grads('conv_1d'),
                lambda geom, num_layers, rng: conv_init_params('conv_1d', geom, num_layers, rng),
                lambda geom, num_layers: conv_num_params('conv_1d', geom, num_layers)),
    'res_net_1d': (_conv_psi('res_net_1d'), _conv_logit('res_net_1d'),
                   _conv_weighted_grads('res_net_1d'),
                   lambda geom, num_layers, rng: conv_init_params('res_net_1d', geom, num_layers, rng),
                   lambda geom, num_layers: conv_num_params('res_net_1d', geom, num_layers)),
}
