// Gradient accumulators, Adam, update_norm, the device-resident epochs and the evaluation loop (training.py:513-778,
// evaluation.py:77-152).  Split out of vmc_api.hip in round 6.
#include "vmc_ctx.hpp"

using namespace vmcapi;

extern "C" {

// sum_b O_k(b) -> g1, sum_b w_b O_k(b) -> g2 for the psi parameter set
// `e` / `mode`: the scalar accumulators (sum E, counts, sum ratio) ride in the reduction launch of the
// dense weight-gradient GEMMs; *scalars_done tells the caller whether they did
static int gradient_sums(vmc_ctx* c, const float* w, bool fresh, const float* e, int mode, bool* scalars_done,
                         bool fold_eloc = false, float beta = 0.f) {
  *scalars_done = false;
  ParamSet& p = c->ps[0];
  const int B = c->B, N = c->N, H = c->H, Hp = c->Hp, NH = c->n_hh;
  float* g1 = c->acc;
  float* g2 = c->acc + c->P;
  Timer t(c, "grad");
  if (c->conv_general) return cgen_gradient_sums(c, w);
  if (c->conv) {
    // forward tapes (the inputs of every convolution), d logit / d (output of every convolution)
    // back through the transposed convolutions, then the weight-gradient correlations
    if (!c->acts_valid) {
      PROPAGATE(conv_rows(c, VMC_PSI, c->configs, c->rowinfo_id, B, nullptr, false, p.logit, true));
      c->acts_valid = true;
    }
    if (c->oact != VMC_ACT_EXP_) HIPCHK(c, launch_out_scale(c->stream, p.logit, c->oscale, B, c->oact));
    ConvBackArgs bk;
    memset(&bk, 0, sizeof(bk));
    bk.g = c->cg; bk.p = conv_params(p); bk.tape = c->ctape; bk.tape_stride = c->ctape_stride;
    bk.oscale = c->oscale; bk.delta = c->cdelta; bk.delta_stride = c->cdelta_stride; bk.B = B; bk.G = c->cG;
    HIPCHK(c, launch_conv_back(c->stream, bk, c->num_cus));
    ConvDwArgs dw;
    memset(&dw, 0, sizeof(dw));
    dw.g = c->cg; dw.configs = c->configs; dw.tape = c->ctape; dw.tape_stride = c->ctape_stride;
    dw.delta = c->cdelta; dw.delta_stride = c->cdelta_stride; dw.w = w; dw.B = B;
    dw.n_slices = c->c_slices; dw.ws = c->cws; dw.g1 = g1; dw.g2 = g2;
    HIPCHK(c, launch_conv_dw(c->stream, dw));
    return VMC_OK;
  }
  // forward with saved activations (wavefunctions.py:345-349 / 418-420); after a sweep launch
  // the kernel has already left them in act[] (exact refresh of the final chains).
  // act[l] = relu(z_{l+1}); RBM: the last one is tanh(z_last) = d sum log cosh / d z_last
  if (!c->acts_valid) {
    if (c->rbm && NH == 0) HIPCHK(c, launch_tanh_copy(c->stream, p.z1, c->act[0], (long long)B * Hp));
    else HIPCHK(c, launch_act_copy(c->stream, p.z1, c->act[0], c->dact_all, (long long)B * Hp, c->hact));
  }
  for (int l = 1; l <= NH && !c->acts_valid; ++l) {
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.A = c->act[l - 1]; g.sam = Hp; g.sak = 1;
    g.B = p.theta + off_w(c, l); g.sbk = H; g.sbn = 1;
    g.M = B; g.N = H; g.K = H; g.C = c->act[l]; g.ldc = Hp;
    g.bias = p.theta + off_b(c, l); g.epilogue = (c->rbm && l == NH) ? 7 : 1; g.splitk = 1;
    g.act = c->hact;
    if (c->dact_all && g.epilogue == 1) g.dact_out = c->dact_all + (long long)l * B * Hp;
    HIPCHK(c, launch_gemm(c->stream, g));
  }
  // psi = g(x) with a non-exp output activation: O_k carries the per-sample factor g'(x) / g(x)
  if (c->oact != VMC_ACT_EXP_) HIPCHK(c, launch_out_scale(c->stream, p.logit, c->oscale, B, c->oact));
  // back-propagation of d logit / d z_l: FC delta[NH] = w_out (.) relu'; RBM delta[NH] = tanh(z)
  // (which IS act[NH]); then the W_l^T chain through the relu masks -- one launch, 16 chains per
  // workgroup, transposed weight fragments on 16x16x4 MFMA (k_backprop16)
  if (c->wide && !c->wide_fast) {
    // delta_NH = w_out (.) f'(z_NH); delta_{l-1} = f'(z_{l-1}) (.) (delta_l W_l^T) on the generic GEMM
    if (c->rbm)   // d sum log cosh(z) / d z = tanh(z), which the forward left in act[NH]
      HIPCHK(c, hipMemcpyAsync(c->delta[NH], c->act[NH], (size_t)B * Hp * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    else
    HIPCHK(c, launch_wide_delta_last(c->stream, c->act[NH], p.woutp, c->oact != VMC_ACT_EXP_ ? c->oscale : nullptr,
                                     B, H, Hp, c->hact, c->delta[NH],
                                     c->dact_all ? c->dact_all + (long long)NH * B * Hp : nullptr));
    for (int l = NH; l >= 1; --l) {
      GemmArgs g; memset(&g, 0, sizeof(g));
      g.A = c->delta[l]; g.sam = Hp; g.sak = 1;
      g.B = p.theta + off_w(c, l); g.sbk = 1; g.sbn = H;          // B(k = out, n = in) = W_l[in][out]
      g.M = B; g.N = H; g.K = H; g.C = c->delta[l - 1]; g.ldc = Hp;
      g.bias = c->wide_zero; g.mask = c->act[l - 1]; g.ldmask = Hp; g.epilogue = 5; g.splitk = 1; g.act = c->hact;
      if (c->dact_all) { g.mask = c->dact_all + (long long)(l - 1) * B * Hp; g.epilogue = 9; }   // cosine: the stored f'(z)
      HIPCHK(c, launch_gemm(c->stream, g));
    }
  } else
  HIPCHK(c, launch_backprop16(c->stream, c->act_all, c->delta_all, p.p16t, p.woutp, B, Hp, NH, c->rbm, c->hact,
                              c->dact_all, c->oact != VMC_ACT_EXP_ ? c->oscale : nullptr,
                              // the fold of the local energies (EnergyGradient: psi's; LogOverlapITSWO: the
                              // supervisor's, and the ratio behind them) rides in this launch
                              !fold_eloc ? ElocFold{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0}
                              : mode == VMC_MODE_ENERGY_GRADIENT
                                  ? ElocFold{c->off, c->diag, c->val, c->offdiag, c->ps[0].eloc, nullptr, nullptr, nullptr, 0.f, 0.f, 0}
                                  : ElocFold{c->off, c->diag, c->val, c->offdiag, c->ps[1].eloc, c->ratio, c->ps[0].logit,
                                             c->ps[1].logit, c->ps[0].shift - c->ps[1].shift, beta, c->oact},
                              OutLayerSums{c->wg_out_partials ? c->wg_outpart : nullptr, w}));
  // Every weight gradient is [a_{l-1} | 1]^T [delta_l | w (.) delta_l]: rows 0..K_in-1 give dW, the
  // implicit ones row gives db (b_l sits right behind w_l in theta), the unscaled product goes to g1 and
  // the w-scaled one to g2.  All NH+2 of them and the scalar accumulators run as
  // ONE launch (k_wgrad); the problem table is built once per weight vector `w`.
  const int slot = (w == c->ratio) ? 1 : 0, par = c->parity;
  if (!c->batch_ready[slot][par]) {
    std::vector<unsigned char> tab((size_t)(NH + 2) * wgrad_problem_bytes(), 0);
    int n = 0, tile0 = 0;
    auto add = [&](const float* a, long long a_ld, int k_in, const float* delta, long long ldd, int n_out, long long off) {
      wgrad_fill_problem(tab.data(), n++, a, a_ld, delta, ldd, off, k_in, n_out, tile0);
      tile0 += plan_wgrad_tiles(k_in, n_out);
    };
    if (c->rbm)   // onsite layer: d logit / d w_on = x, d logit / d b_on = 1
      add(c->configs, N, N, c->ones, 1, 1, c->lay.off_won);
    else if (!c->wg_out_partials)   // output layer: d logit / d w_out = a_L, d logit / d b_out = 1
      add(c->act[NH], Hp, H, c->oscale, 1, 1, off_wout(c));   // oscale == 1 for the exp output
    for (int l = NH; l > 0; --l) add(c->act[l - 1], Hp, H, c->delta[l], Hp, H, off_w(c, l));
    add(c->configs, N, N, c->delta[0], Hp, H, off_w(c, 0));
    if (tile0 != c->wg_tiles) return fail(c, VMC_ERR_STATE, "weight-gradient tile count does not match the plan");
    HIPCHK(c, hipMemcpy(c->d_batch[slot][par], tab.data(), tab.size(), hipMemcpyHostToDevice));
    c->batch_ready[slot][par] = true;
  }
  {
    const char* fe = getenv("CGS_VMC_WGRAD_SLICES");        // measurement / test knob, read per launch
    const int forced = fe ? atoi(fe) : 0;
    WgradLaunch L;
    memset((void*)&L, 0, sizeof(L));
    L.dev_problems = c->d_batch[slot][par]; L.n_prob = NH + (c->wg_out_partials ? 1 : 2);
    if (c->wg_out_partials) {
      L.out_part = c->wg_outpart; L.out_nwg = (B + 15) / 16; L.out_H = H; L.out_ld = Hp + 4; L.out_off = off_wout(c);
    }
    L.tiles = c->wg_tiles;
    L.slices = plan_wgrad_slices(c->wg_tiles, B, c->num_cus, 1 + (c->wg_out_partials ? plan_wgrad_fold_blocks(H) : 0), forced);
    L.K = B; L.w = w; L.g1 = g1; L.g2 = g2; L.ws = c->gemm_ws; L.tickets = c->wg_tickets; L.fresh = fresh;
    L.sc_eloc = e; L.sc_ratio = mode == 1 ? c->ratio : nullptr; L.sc_out = c->acc + 2 * c->P; L.sc_B = B; L.sc_mode = mode;
    HIPCHK(c, launch_wgrad(c->stream, L));
  }
  *scalars_done = true;
  return VMC_OK;
}

// SR sample store: the chains of this accumulate call with their activations a_l and
// back-propagated d logit / d z_l, which gradient_sums has just left in act[] / delta[]
static int sr_record(vmc_ctx* c) {
  if (c->sr_n >= c->sr_cap)
    return fail(c, VMC_ERR_STATE, "SR sample store full: vmc_sr_reserve fewer batches than accumulate calls");
  const long long B = c->B, N = c->N, Hp = c->Hp, L = c->A, k = c->sr_n, R = (long long)c->sr_cap * B;
  HIPCHK(c, hipMemcpyAsync(c->sr_cfg + k * B * N, c->configs, B * N * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  if (c->conv_general) { c->sr_n += 1; return VMC_OK; }     // (its matvec re-derives everything from the chains)
  if (c->conv) {   // the taped inputs of every convolution and d logit / d (their outputs) of this batch
    const long long CS = c->cg.CS, nc = c->cg.n_conv;
    if (nc > 1)
      HIPCHK(c, hipMemcpy2DAsync(c->sr_ctape + k * B * CS, R * CS * sizeof(float), c->ctape, c->ctape_stride * sizeof(float),
                                 B * CS * sizeof(float), nc - 1, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpy2DAsync(c->sr_cdelta + k * B * CS, R * CS * sizeof(float), c->cdelta, c->cdelta_stride * sizeof(float),
                               B * CS * sizeof(float), nc, hipMemcpyDeviceToDevice, c->stream));
    c->sr_n += 1;
    return VMC_OK;
  }
  // layer-major store [L][cap * B][Hp]: every layer's rows of ALL stored batches are contiguous,
  // so the CG matrix-vector product runs each GEMM once over all samples
  HIPCHK(c, hipMemcpy2DAsync(c->sr_act + k * B * Hp, R * Hp * sizeof(float), c->act_all, B * Hp * sizeof(float),
                             B * Hp * sizeof(float), L, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpy2DAsync(c->sr_delta + k * B * Hp, R * Hp * sizeof(float), c->delta_all, B * Hp * sizeof(float),
                             B * Hp * sizeof(float), L, hipMemcpyDeviceToDevice, c->stream));
  c->sr_n += 1;
  return VMC_OK;
}

int vmc_accumulate(vmc_ctx* c, int mode, float beta) {
  ENTER(c);
  if (mode != VMC_MODE_ENERGY_GRADIENT && mode != VMC_MODE_LOG_OVERLAP_ITSWO)
    return fail(c, VMC_ERR_INVALID, "bad mode");
  const float* w = nullptr;
  const float* e = nullptr;
  // everything this call enqueues comes after ev_mark; a sampler launch that follows directly
  // may start as soon as ev_mark has passed (vmc_mc_steps).  If this call has to rebuild the
  // psi cache the sampler reads, the launch must wait for all of it instead.
  const bool cache_was_valid = c->ps[0].cache_valid && c->ps[0].packed_valid;
  if (can_overlap(c)) HIPCHK(c, hipEventRecord(c->ev_mark, c->stream));
  bool fold_eloc = false, refresh_ran = false;
  if (mode == VMC_MODE_ENERGY_GRADIENT) {
    PROPAGATE(local_energy_device(c, VMC_PSI, true, &fold_eloc));   // training.py:542-543
    w = e = c->ps[0].eloc;
  } else {
    if (!c->ps[1].has_params) return fail(c, VMC_ERR_STATE, "supervisor parameters not set (vmc_transfer_params)");
    if (!c->ps[1].cache_valid && sampler_refresh_ok(c)) {
      // the refresh pass is a sampler launch: it writes its chain copy to configs_alt, the buffer a directly
      // following vmc_mc_steps writes too.  Behind ev_mark alone that launch could overtake it and have its
      // new chains overwritten by the refresh's old ones: no token, the sampler waits for all of this.
      PROPAGATE(refresh_cache_by_sampler(c, VMC_OMEGA));
      refresh_ran = true;
    }
    PROPAGATE(local_energy_device(c, VMC_OMEGA, true, &fold_eloc));   // training.py:664, 667
    PROPAGATE(ensure_cache(c, VMC_PSI));
    if (!fold_eloc)   // (otherwise the back-propagation launch folds E_loc^w and forms the ratio: two launches less)
      HIPCHK(c, launch_itswo_ratio(c->stream, c->ps[0].logit, c->ps[1].logit, c->ps[1].eloc,
                                   c->ps[0].shift - c->ps[1].shift, beta, c->B, c->ratio, c->oact));
    w = c->ratio; e = c->ps[1].eloc;
  }
  PROPAGATE(ensure_cache(c, VMC_PSI));
  // the batched weight-gradient GEMMs of the dense ansatz types cover every parameter, so a pending
  // reset is absorbed: their reduction stores instead of adding (conv: zero first)
  if (c->conv) PROPAGATE(acc_zeros(c));
  const bool fresh = c->acc_fresh;
  bool scalars_done = false;
  PROPAGATE(gradient_sums(c, w, fresh, e, mode, &scalars_done, fold_eloc, beta));
  if (!scalars_done)
    HIPCHK(c, launch_scalar_accum(c->stream, e, mode == 1 ? c->ratio : nullptr, c->B, c->acc + 2 * c->P, mode, fresh));
  c->acc_fresh = false;
  if (c->sr_cap > 0 && mode == VMC_MODE_ENERGY_GRADIENT) PROPAGATE(sr_record(c));
  c->acc_since_sweep = true;
  c->token = cache_was_valid && !refresh_ran;
  return VMC_OK;
}

int vmc_reset_accumulators(vmc_ctx* c) {
  CHECK_CTX(c);
  c->acc_fresh = true;            // zeroed lazily: see vmc_ctx::acc_fresh
  c->sr_n = 0; c->sr_begun = false;
  return VMC_OK;
}

int vmc_get_accumulators(vmc_ctx* c, float* host) {
  CHECK_CTX(c);
  PROPAGATE(acc_zeros(c));
  if (!host) return fail(c, VMC_ERR_INVALID, "null");
  HIPCHK(c, hipMemcpyAsync(host, c->acc, (2 * c->P + 8) * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return VMC_OK;
}

int vmc_set_accumulators(vmc_ctx* c, const float* host) {
  CHECK_CTX(c);
  if (!host) return fail(c, VMC_ERR_INVALID, "null");
  c->acc_fresh = false;           // fully overwritten
  HIPCHK(c, hipMemcpyAsync(c->acc, host, (2 * c->P + 8) * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return VMC_OK;
}

int vmc_mean_energy(vmc_ctx* c, double* energy) {
  CHECK_CTX(c);
  PROPAGATE(acc_zeros(c));
  if (!energy) return fail(c, VMC_ERR_INVALID, "null");
  float sc[8];
  HIPCHK(c, hipMemcpyAsync(sc, c->acc + 2 * c->P, 8 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *energy = (double)(sc[0] / sc[1]);   // tf.metrics.mean value: total / count
  return VMC_OK;
}

int vmc_get_gradient(vmc_ctx* c, int mode, float* grad) {
  CHECK_CTX(c);
  if (!grad || (mode != 0 && mode != 1)) return fail(c, VMC_ERR_INVALID, "bad arguments");
  PROPAGATE(acc_zeros(c));
  HIPCHK(c, launch_adam(c->stream, nullptr, nullptr, nullptr, c->acc, (int)c->P, mode, 0.f, 0.f, 0.f, 0.f, c->grad_tmp));
  HIPCHK(c, hipMemcpyAsync(grad, c->grad_tmp, c->P * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return VMC_OK;
}

int vmc_apply_adam(vmc_ctx* c, int mode, float lr, float beta1, float beta2, float eps, double* energy) {
  ENTER(c);
  if (mode != 0 && mode != 1) return fail(c, VMC_ERR_INVALID, "bad mode");
  if (!c->ps[0].has_params) return fail(c, VMC_ERR_STATE, "parameters not set");
  PROPAGATE(acc_zeros(c));
  c->adam_t += 1;
  const float t = (float)c->adam_t;
  const float lr_t = lr * sqrtf(1.f - powf(beta2, t)) / (1.f - powf(beta1, t));
  {
    Timer tm(c, "adam");
    HIPCHK(c, launch_adam(c->stream, c->ps[0].theta, c->adam_m, c->adam_v, c->acc, (int)c->P, mode, lr_t, beta1, beta2, eps, nullptr));
  }
  c->ps[0].packed_valid = c->ps[0].cache_valid = false;
  c->acts_valid = false;
  if (energy) PROPAGATE(vmc_mean_energy(c, energy));
  return VMC_OK;
}

int vmc_get_adam_state(vmc_ctx* c, float* m, float* v, int64_t* t) {
  CHECK_CTX(c);
  if (m) HIPCHK(c, hipMemcpyAsync(m, c->adam_m, c->P * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (v) HIPCHK(c, hipMemcpyAsync(v, c->adam_v, c->P * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (t) *t = c->adam_t;
  return VMC_OK;
}

int vmc_set_adam_state(vmc_ctx* c, const float* m, const float* v, int64_t t) {
  CHECK_CTX(c);
  if (m) HIPCHK(c, hipMemcpyAsync(c->adam_m, m, c->P * sizeof(float), hipMemcpyHostToDevice, c->stream));
  if (v) HIPCHK(c, hipMemcpyAsync(c->adam_v, v, c->P * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->adam_t = t;
  return VMC_OK;
}

// Wavefunction.update_norm (wavefunctions.py:261-288); max_b psi over the chains of all ranks
static int update_norm_impl(vmc_ctx* c, void* comm, int world, float max_value) {
  if (c->oact != VMC_ACT_EXP_) return VMC_OK;   // wavefunctions.py:276-277: no exp_norm_shift, nothing to do
  PROPAGATE(ensure_cache(c, VMC_PSI));
  HIPCHK(c, launch_max(c->stream, c->ps[0].logit, c->B, c->d_max));
  PROPAGATE(reduce_buffer(c, comm, world, c->d_max, 1, VMC_REDUCE_MAX));
  float mx = 0.f;
  HIPCHK(c, hipMemcpyAsync(&mx, c->d_max, sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // wavefunctions.py:280-288: log_max = log(reduce_max(psi)); where psi overflows float32 the
  // reference yields inf; the logit-domain value is used there instead.
  const float shift = c->ps[0].shift;
  const float psi_max = expf(mx - shift);
  const float log_max = (std::isfinite(psi_max) && psi_max > 0.f) ? logf(psi_max) : (mx - shift);
  const float max_log = logf(max_value);
  if (log_max > max_log) c->ps[0].shift = shift + (log_max - max_log);
  return VMC_OK;
}

int vmc_update_norm(vmc_ctx* c, float max_value) {
  ENTER(c);
  return update_norm_impl(c, nullptr, 1, max_value);
}

int vmc_update_norm_dist(vmc_ctx* c, void* nccl_comm, int32_t world_size, float max_value) {
  ENTER(c);
  return update_norm_impl(c, nccl_comm, world_size, max_value);
}

static bool side_sweep_enabled() { const char* e = getenv("CGS_VMC_SIDE_SWEEP"); return !(e && atoi(e) == 0); }

static int epoch_energy_gradient_impl(vmc_ctx* c, void* comm, int world, int64_t n_eq_steps, int32_t n_batches,
                                      int64_t n_mc_steps, float max_value) {
  if (n_eq_steps < 0 || n_batches < 0 || n_mc_steps < 0) return fail(c, VMC_ERR_INVALID, "negative count");
  PROPAGATE(vmc_mc_steps(c, n_eq_steps, nullptr));                       // training.py:608-609
  if (max_value > 0.f) {                                                  // training.py:611-612
    PROPAGATE(join_sweep(c));
    PROPAGATE(update_norm_impl(c, comm, world, max_value));
  }
  PROPAGATE(vmc_reset_accumulators(c));                                   // training.py:613
  for (int b = 0; b < n_batches; ++b) {                                   // training.py:614-617
    c->expect_sweep = n_mc_steps > 0;
    PROPAGATE(vmc_accumulate(c, VMC_MODE_ENERGY_GRADIENT, 0.f));
    // sharded chains: the last sweep does not touch the accumulators -- on its own stream it runs beside the
    // all-reduce instead of in front of it (CGS_VMC_SIDE_SWEEP=0: in stream order, for A/B)
    if (b == n_batches - 1 && world > 1 && n_mc_steps > 0 && side_sweep_enabled()) c->side_sweep_once = true;
    PROPAGATE(vmc_mc_steps(c, n_mc_steps, nullptr));
  }
  // sharded chains: the accumulators leave this call summed over ranks (the last sweep, on its own
  // stream, keeps running underneath the collective)
  PROPAGATE(reduce_accumulators(c, comm, world));
  return VMC_OK;
}

int vmc_epoch_energy_gradient(vmc_ctx* c, int64_t n_eq_steps, int32_t n_batches, int64_t n_mc_steps,
                              float max_value) {
  ENTER(c);
  return epoch_energy_gradient_impl(c, nullptr, 1, n_eq_steps, n_batches, n_mc_steps, max_value);
}

int vmc_epoch_energy_gradient_dist(vmc_ctx* c, void* nccl_comm, int32_t world_size, int64_t n_eq_steps,
                                   int32_t n_batches, int64_t n_mc_steps, float max_value) {
  ENTER(c);
  return epoch_energy_gradient_impl(c, nccl_comm, world_size, n_eq_steps, n_batches, n_mc_steps, max_value);
}

static int epoch_log_overlap_impl(vmc_ctx* c, void* comm, int world, float beta, int64_t n_eq_steps,
                                  int32_t n_batches, int64_t n_mc_steps, float max_value, float lr,
                                  float beta1, float beta2, float eps, double* energy) {
  if (n_eq_steps < 0 || n_batches < 0 || n_mc_steps < 0) return fail(c, VMC_ERR_INVALID, "negative count");
  PROPAGATE(vmc_mc_steps(c, n_eq_steps, nullptr));                       // training.py:750-751
  if (max_value > 0.f) {                                                  // training.py:753-754
    PROPAGATE(join_sweep(c));
    PROPAGATE(update_norm_impl(c, comm, world, max_value));
  }
  PROPAGATE(vmc_transfer_params(c));                                      // training.py:755
  for (int b = 0; b < n_batches; ++b) {                                   // training.py:756-761
    PROPAGATE(vmc_mc_steps(c, n_mc_steps, nullptr));
    PROPAGATE(vmc_reset_accumulators(c));
    PROPAGATE(vmc_accumulate(c, VMC_MODE_LOG_OVERLAP_ITSWO, beta));
    PROPAGATE(reduce_accumulators(c, comm, world));   // in stream: Adam sees the sums over all ranks
    PROPAGATE(vmc_apply_adam(c, VMC_MODE_LOG_OVERLAP_ITSWO, lr, beta1, beta2, eps, nullptr));
  }
  if (energy) PROPAGATE(vmc_mean_energy(c, energy));                      // training.py:763
  return VMC_OK;
}

int vmc_epoch_log_overlap(vmc_ctx* c, float beta, int64_t n_eq_steps, int32_t n_batches,
                          int64_t n_mc_steps, float max_value, float lr, float beta1, float beta2,
                          float eps, double* energy) {
  ENTER(c);
  return epoch_log_overlap_impl(c, nullptr, 1, beta, n_eq_steps, n_batches, n_mc_steps, max_value, lr, beta1, beta2,
                                eps, energy);
}

int vmc_epoch_log_overlap_dist(vmc_ctx* c, void* nccl_comm, int32_t world_size, float beta, int64_t n_eq_steps,
                               int32_t n_batches, int64_t n_mc_steps, float max_value, float lr, float beta1,
                               float beta2, float eps, double* energy) {
  ENTER(c);
  return epoch_log_overlap_impl(c, nccl_comm, world_size, beta, n_eq_steps, n_batches, n_mc_steps, max_value, lr,
                                beta1, beta2, eps, energy);
}

// MonteCarloOperatorEvaluator.run_evaluation (evaluation.py:138-145) without a host round trip per
// sample: the batch sums go to d_eval[s]; one float64 all-reduce of the per-rank means at the end.
int vmc_evaluate(vmc_ctx* c, void* nccl_comm, int32_t world_size, int64_t n_eq_steps, int32_t n_samples,
                 int64_t n_mc_steps, double* means, int64_t* accepted) {
  ENTER(c);
  if (n_eq_steps < 0 || n_samples < 0 || n_mc_steps < 0) return fail(c, VMC_ERR_INVALID, "negative count");
  if (n_samples > 0 && !means) return fail(c, VMC_ERR_INVALID, "null means");
  if (c->n_bonds <= 0) return fail(c, VMC_ERR_STATE, "bonds not set (vmc_set_bonds)");
  if (n_samples > c->d_eval_n) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->d_eval) hipFree(c->d_eval);
    c->d_eval = nullptr; c->d_eval_n = 0;
    HIPCHK(c, dalloc(&c->d_eval, n_samples));
    c->d_eval_n = n_samples;
  }
  PROPAGATE(vmc_mc_steps(c, n_eq_steps, nullptr));                        // evaluation.py:135-136
  PROPAGATE(join_sweep(c));
  // the samplers add their acceptances to the device counter; it is read once, at the end
  HIPCHK(c, hipMemsetAsync(c->d_accepted, 0, sizeof(unsigned long long), c->stream));
  for (int s = 0; s < n_samples; ++s) {                                   // evaluation.py:138-145
    PROPAGATE(join_sweep(c));
    PROPAGATE(local_energy_device(c, VMC_PSI));
    HIPCHK(c, launch_sum(c->stream, c->ps[0].eloc, c->B, c->d_eval + s));
    PROPAGATE(vmc_mc_steps(c, n_mc_steps, nullptr));
  }
  PROPAGATE(join_sweep(c));
  const int world = world_size > 1 ? world_size : 1;
  if (n_samples > 0) {
    HIPCHK(c, launch_div_f64(c->stream, c->d_eval, n_samples, (double)c->B));   // this rank's batch means
    if (sharded(nccl_comm, world_size))
      PROPAGATE(reduce_buffer(c, nccl_comm, world_size, c->d_eval, n_samples, VMC_REDUCE_SUM_F64));
    HIPCHK(c, hipMemcpyAsync(means, c->d_eval, n_samples * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  }
  unsigned long long h_acc = 0;
  int cnt_total = 0;
  HIPCHK(c, hipMemcpyAsync(&h_acc, c->d_accepted, sizeof(h_acc), hipMemcpyDeviceToHost, c->stream));
  if (n_samples > 0) HIPCHK(c, hipMemcpyAsync(&cnt_total, c->off + c->B, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (n_samples > 0) c->last_rows = cnt_total;
  if (world > 1)
    for (int s = 0; s < n_samples; ++s) means[s] /= (double)world;       // mean over ALL ranks' chains
  if (accepted) *accepted = (int64_t)h_acc;
  return VMC_OK;
}


}  // extern "C"
