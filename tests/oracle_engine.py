"""Test double: the subset of cgs_vmc_amd.engine.VmcEngine that training.py / graph_builders.py /
wavefunctions.py drive, backed by the numpy oracle (fp64) instead of libcgsvmc_hip.so.

TEST INFRASTRUCTURE ONLY.  It lets the world_size-2 gloo test (no GPU) run the product's Python
routing for sharded chains -- training.run_optimization_epoch -> engine.epoch_*_dist(collective)
-> Collective.allreduce_host -- and compare it with the unsharded oracle epoch.  The `_dist`
methods issue their all-reduces at exactly the points where csrc/vmc_api_train.hip / vmc_api_sr.hip
(epoch_energy_gradient_impl / epoch_log_overlap_impl / sr_solve_impl) issues them.
fully_connected ansatz only."""
import numpy as np

from oracle import vmc_oracle as vo


class OracleEngine:

  def __init__(self, n_sites, batch_size, num_layers, layer_size, nonlinearity='relu',
               output_activation='exp', device=0, chain_offset=0, seed=2024, stream=0,
               ansatz='fully_connected', kernel_size=0, size_x=0, size_y=0):
    assert ansatz == 'fully_connected'
    self.n_sites, self.batch_size = n_sites, batch_size
    self.num_layers, self.layer_size = num_layers, layer_size
    self.nonlinearity, self.output_activation = nonlinearity, output_activation
    self.chain_offset, self.seed, self.device, self.ansatz = chain_offset, seed, device, ansatz
    self.num_params = vo.num_params(n_sites, layer_size, num_layers)
    self.theta = [None, None]
    self.shift = [np.float32(-10.0), np.float32(-10.0)]
    self.configs = None
    self.step = 0
    self.acc = vo.Accumulators(self.num_params, np.float64)
    self.adam = vo.AdamState(self.num_params)
    self.sr_cap = 0
    self.samples = []
    self.n_bonds = 0

  def close(self):
    pass

  # ---- state
  def set_bonds(self, bonds, j_x, j_z):
    jx, jz = np.unique(np.asarray(j_x, np.float64)), np.unique(np.asarray(j_z, np.float64))
    assert jx.size == 1 and jz.size == 1, 'the double takes uniform couplings'
    self.bonds, self.j_x, self.j_z = [tuple(int(i) for i in b) for b in bonds], float(jx[0]), float(jz[0])
    self.n_bonds = len(self.bonds)

  def set_params(self, theta, which=0):
    self.theta[which] = np.ascontiguousarray(theta, np.float32).copy()

  def get_params(self, which=0):
    return self.theta[which].copy()

  def transfer_params(self):
    self.theta[1] = self.theta[0].copy()

  def set_configs(self, configs):
    if np.shape(configs) != (self.batch_size, self.n_sites):
      raise ValueError('Size of existing variable does not match.')
    self.configs = np.ascontiguousarray(configs, np.float32).copy()

  def get_configs(self):
    return self.configs.copy()

  def set_shift(self, shift, which=0):
    self.shift[which] = np.float32(shift)

  def get_shift(self, which=0):
    return float(self.shift[which])

  # ---- hot path
  def _kw(self):
    return dict(nonlinearity=self.nonlinearity, output_activation=self.output_activation)

  def _logit(self, which=0):
    return vo.fc_logit(self.theta[which], self.configs, self.layer_size, self.num_layers,
                       self.nonlinearity, np.float64)

  def mc_steps(self, n_steps, want_accepted=True):
    self.configs, acc = vo.run_sweeps(self.theta[0], self.configs, int(n_steps), self.seed,
                                      self.step, self.layer_size, self.num_layers,
                                      shift=float(self.shift[0]), chain_offset=self.chain_offset,
                                      dtype=np.float64, **self._kw())
    self.configs = self.configs.astype(np.float32)
    self.step += int(n_steps)
    return acc

  def reset_accumulators(self):
    self.acc.reset()
    self.samples = []

  def accumulate(self, mode, beta=0.0):
    if mode == 0:
      vo.energy_gradient_accumulate(self.acc, self.theta[0], self.configs, self.bonds, self.j_x,
                                    self.j_z, float(self.shift[0]), self.layer_size,
                                    self.num_layers, np.float64, **self._kw())
      if self.sr_cap:
        self.samples.append(self.configs.copy())
    else:
      vo.log_overlap_accumulate(self.acc, self.theta[0], self.theta[1], self.configs, self.bonds,
                                self.j_x, self.j_z, float(self.shift[0]), float(self.shift[1]),
                                beta, self.layer_size, self.num_layers, np.float64, **self._kw())

  def _pack(self):
    a = self.acc
    return np.concatenate([a.g1_total, a.g2_total,
                           [a.e_total, a.e_count, a.r_total, a.r_count, a.g_count, 0, 0, 0]])

  def _unpack(self, buf):
    p, a = self.num_params, self.acc
    a.g1_total, a.g2_total = buf[:p].copy(), buf[p:2 * p].copy()
    a.e_total, a.e_count, a.r_total, a.r_count, a.g_count = buf[2 * p:2 * p + 5]

  def get_accumulators(self):
    return self._pack().astype(np.float32)

  def _allreduce_accumulators(self, coll):
    # fp64 payload in an fp32 wire format would lose the point of the fp64 twin: reduce hi + lo
    buf = self._pack()
    hi = buf.astype(np.float32)
    lo = (buf - hi).astype(np.float32)
    coll.allreduce_host(hi, 'sum'); coll.allreduce_host(lo, 'sum')
    red = hi.astype(np.float64) + lo.astype(np.float64)
    red[2 * self.num_params + 4] /= coll.world       # g_count: calls, not calls x ranks
    self._unpack(red)

  def allreduce_accumulators_dist(self, coll):
    self._allreduce_accumulators(coll)

  def apply_adam(self, mode, lr, beta1=0.9, beta2=0.99, eps=1e-8):
    grad = vo.energy_gradient(self.acc) if mode == 0 else vo.log_overlap_gradient(self.acc)
    self.theta[0] = vo.adam_apply(self.adam, self.theta[0], grad, lr, beta1, beta2, eps)
    return float(self.acc.mean_energy())

  def mean_energy(self):
    return float(self.acc.mean_energy())

  def _update_norm(self, coll, max_value):
    top = np.array([self._logit(0).max()], np.float32)
    if coll is not None:
      coll.allreduce_host(top, 'max')
    gap = np.float32(top[0] - self.shift[0])
    max_log = np.log(np.float32(max_value))
    if gap > max_log:
      self.shift[0] = np.float32(self.shift[0] + (gap - max_log))

  def update_norm(self, max_value=1e10):
    self._update_norm(None, max_value)

  def update_norm_dist(self, coll, max_value=1e10):
    self._update_norm(coll, max_value)

  # ---- whole-epoch entries (single rank and sharded)
  def epoch_energy_gradient(self, n_eq, n_batches, n_mc, max_value=1e10):
    self._epoch_eg(None, n_eq, n_batches, n_mc, max_value)

  def epoch_energy_gradient_dist(self, coll, n_eq, n_batches, n_mc, max_value=1e10):
    self._epoch_eg(coll, n_eq, n_batches, n_mc, max_value)

  def _epoch_eg(self, coll, n_eq, n_batches, n_mc, max_value):
    self.mc_steps(n_eq)
    if max_value > 0:
      self._update_norm(coll, max_value)
    self.reset_accumulators()
    for _ in range(n_batches):
      self.accumulate(0)
      self.mc_steps(n_mc)
    if coll is not None:
      self._allreduce_accumulators(coll)

  def epoch_log_overlap(self, beta, n_eq, n_batches, n_mc, max_value, lr, beta1, beta2, eps):
    return self._epoch_lo(None, beta, n_eq, n_batches, n_mc, max_value, lr, beta1, beta2, eps)

  def epoch_log_overlap_dist(self, coll, beta, n_eq, n_batches, n_mc, max_value, lr, beta1,
                             beta2, eps):
    return self._epoch_lo(coll, beta, n_eq, n_batches, n_mc, max_value, lr, beta1, beta2, eps)

  def _epoch_lo(self, coll, beta, n_eq, n_batches, n_mc, max_value, lr, beta1, beta2, eps):
    self.mc_steps(n_eq)
    if max_value > 0:
      self._update_norm(coll, max_value)
    self.transfer_params()
    for _ in range(n_batches):
      self.mc_steps(n_mc)
      self.reset_accumulators()
      self.accumulate(1, beta)
      if coll is not None:
        self._allreduce_accumulators(coll)
      self.apply_adam(1, lr, beta1, beta2, eps)
    return self.mean_energy()

  # ---- stochastic reconfiguration (extension): explicit per-sample O, CG as in csrc/sr.hip
  def sr_reserve(self, n_batches):
    self.sr_cap = int(n_batches)

  def _sr_cg(self, coll, diag_shift, tol, max_iter):
    a, p_ = self.acc, self.num_params
    n = float(a.e_count)
    f = a.g2_total / n - (a.e_total / n) * (a.g1_total / n)
    o_mean = a.g1_total / n
    o = vo.per_sample_logit_grads(self.theta[0], np.concatenate(self.samples), self.layer_size,
                                  self.num_layers, self.nonlinearity, np.float64)
    x = np.zeros(p_); r = f.copy(); p = f.copy()
    rr0 = rr = float(r @ r)
    it = 0
    while it < max_iter and rr0 > 0 and rr > tol * tol * rr0:
      t = o @ p
      u = np.concatenate([o.T @ t, [t.sum()]])
      if coll is not None:
        hi = u.astype(np.float32); lo = (u - hi).astype(np.float32)
        coll.allreduce_host(hi, 'sum'); coll.allreduce_host(lo, 'sum')
        u = hi.astype(np.float64) + lo.astype(np.float64)
      q = u[:-1] / n - o_mean * (u[-1] / n) + diag_shift * p
      alpha = rr / float(p @ q)
      x += alpha * p
      r -= alpha * q
      rr_new = float(r @ r)
      p = r + (rr_new / rr) * p
      rr = rr_new
      it += 1
    self.sr_x = x
    return it, (float(np.sqrt(rr / rr0)) if rr0 > 0 else 0.0)

  def sr_solve(self, diag_shift, tol, max_iter):
    return self._sr_cg(None, diag_shift, tol, max_iter)

  def sr_solve_dist(self, coll, diag_shift, tol, max_iter):
    return self._sr_cg(coll, diag_shift, tol, max_iter)

  def sr_apply(self, lr):
    self.theta[0] = (self.theta[0] - np.float32(lr) * self.sr_x).astype(np.float32)
    return self.mean_energy()

  # ---- tensors the op handles may read
  def amplitude(self, configs=None, which=0):
    cfg = self.configs if configs is None else np.asarray(configs, np.float32)
    logit = vo.fc_logit(self.theta[which], cfg, self.layer_size, self.num_layers,
                        self.nonlinearity, np.float64)
    with np.errstate(over='ignore'):
      return logit.astype(np.float32), np.exp(logit - self.shift[which]).astype(np.float32)

  def local_energy(self, which=0, want_eloc=True):
    amp = lambda c: vo.fc_psi(self.theta[which], c, self.layer_size, self.num_layers,
                              float(self.shift[which]), dtype=np.float64, **self._kw())
    e = vo.local_value(amp, self.configs, self.bonds, self.j_x, self.j_z, dtype=np.float64)
    return (e.astype(np.float32) if want_eloc else None), float(e.mean())
