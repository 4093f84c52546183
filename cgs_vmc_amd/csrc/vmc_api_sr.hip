// Stochastic reconfiguration (extension named by the north star; no reference line): the sample store, the matrix-free
// CG entries and the one-call solves.  Split out of vmc_api.hip in round 6.
#include "vmc_ctx.hpp"

using namespace vmcapi;

extern "C" {

// ------------------------------------------------------------------ stochastic reconfiguration
int vmc_sr_reserve(vmc_ctx* c, int32_t n_batches) {
  ENTER(c);
  if (n_batches < 0) return fail(c, VMC_ERR_INVALID, "n_batches < 0");
  if (n_batches > 0 && c->oact != VMC_ACT_EXP_)
    return fail(c, VMC_ERR_UNSUPPORTED, "stochastic reconfiguration (an extension) covers the exp output activation (every hidden activation)");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  void* old[] = {c->sr_cfg, c->sr_act, c->sr_delta, c->sr_ws, c->sr_t, c->sr_ones, c->sr_ctape, c->sr_cdelta, c->sr_cws, c->sr_tpart};
  for (void* q : old) if (q) hipFree(q);
  c->sr_cfg = c->sr_act = c->sr_delta = c->sr_ws = c->sr_t = c->sr_ones = c->sr_tpart = nullptr;
  c->sr_ctape = c->sr_cdelta = c->sr_cws = nullptr;
  c->sr_cap = 0; c->sr_n = 0; c->sr_begun = false;
  if (n_batches == 0) return VMC_OK;
  const long long B = c->B, N = c->N, Hp = c->Hp, L = c->A, P = c->P, R = (long long)n_batches * B;
  if (c->conv_general) {       // the chains are all that is stored (cgen_sr_matvec; single-rank solves only)
    if (R * N >= (1LL << 31)) return fail(c, VMC_ERR_UNSUPPORTED, "SR sample store too large (rows * sites >= 2^31)");
    HIPCHK(c, dalloc(&c->sr_cfg, R * N));
    HIPCHK(c, dalloc(&c->sr_t, R));
    if (!c->sr_u) {
      HIPCHK(c, dalloc(&c->sr_u, P + 1)); HIPCHK(c, dalloc(&c->sr_x, P)); HIPCHK(c, dalloc(&c->sr_r, P));
      HIPCHK(c, dalloc(&c->sr_p, P)); HIPCHK(c, dalloc(&c->sr_q, P));
      HIPCHK(c, dalloc(&c->sr_partial, 256)); HIPCHK(c, dalloc(&c->sr_sc, 4));
      HIPCHK(c, hipMemsetAsync(c->sr_x, 0, P * sizeof(float), c->stream));
    }
    c->sr_cap = n_batches;
    return VMC_OK;
  }
  if (c->conv) {
    const ConvGeom& cg = c->cg;
    const long long CS = cg.CS, nc = cg.n_conv, nl = nc > 1 ? nc - 1 : 1;
    if (R * CS >= (1LL << 31)) return fail(c, VMC_ERR_UNSUPPORTED, "SR sample store too large (rows * feature-map size >= 2^31)");
    HIPCHK(c, dalloc(&c->sr_cfg, R * N));
    HIPCHK(c, dalloc(&c->sr_ctape, nl * R * CS)); HIPCHK(c, dalloc(&c->sr_cdelta, nc * R * CS));
    HIPCHK(c, dalloc(&c->sr_t, R));
    c->sr_cslices = R < 256 ? (int)R : 256;
    HIPCHK(c, dalloc(&c->sr_cws, plan_conv_dw_ws_floats(c->cg, c->sr_cslices)));
    if (!c->sr_cw0) {
      HIPCHK(c, dalloc(&c->sr_cw0, plan_conv_w0_floats(cg))); HIPCHK(c, dalloc(&c->sr_cwf, plan_conv_wf_floats(cg)));
      HIPCHK(c, dalloc(&c->sr_cwb, plan_conv_wf_floats(cg))); HIPCHK(c, dalloc(&c->sr_cbias, plan_conv_bias_floats(cg)));
    }
    if (!c->sr_u) {
      HIPCHK(c, dalloc(&c->sr_u, P + 1)); HIPCHK(c, dalloc(&c->sr_x, P)); HIPCHK(c, dalloc(&c->sr_r, P));
      HIPCHK(c, dalloc(&c->sr_p, P)); HIPCHK(c, dalloc(&c->sr_q, P));
      HIPCHK(c, dalloc(&c->sr_partial, 256)); HIPCHK(c, dalloc(&c->sr_sc, 4));
      HIPCHK(c, hipMemsetAsync(c->sr_x, 0, P * sizeof(float), c->stream));
    }
    c->sr_cap = n_batches;
    return VMC_OK;
  }
  if (R > 0x7fffffffLL / Hp) return fail(c, VMC_ERR_UNSUPPORTED, "SR sample store too large (rows * Hp >= 2^31)");
  HIPCHK(c, dalloc(&c->sr_cfg, R * N));
  HIPCHK(c, dalloc(&c->sr_act, L * R * Hp));
  HIPCHK(c, dalloc(&c->sr_delta, L * R * Hp));
  HIPCHK(c, dalloc(&c->sr_ws, (long long)sr_wsum_slices((int)R, c->num_cus) * ((N > c->H ? N : c->H) + 1) * c->H));
  HIPCHK(c, dalloc(&c->sr_t, R)); HIPCHK(c, dalloc(&c->sr_ones, R));
  HIPCHK(c, dalloc(&c->sr_tpart, L * ((c->H + 255) / 256) * R));
  HIPCHK(c, launch_fill(c->stream, c->sr_ones, 1.f, R));
  if (!c->sr_u) {
    HIPCHK(c, dalloc(&c->sr_u, P + 1)); HIPCHK(c, dalloc(&c->sr_x, P)); HIPCHK(c, dalloc(&c->sr_r, P));
    HIPCHK(c, dalloc(&c->sr_p, P)); HIPCHK(c, dalloc(&c->sr_q, P));
    HIPCHK(c, dalloc(&c->sr_partial, 256)); HIPCHK(c, dalloc(&c->sr_sc, 4));
    HIPCHK(c, hipMemsetAsync(c->sr_x, 0, P * sizeof(float), c->stream));
  }
  c->sr_cap = n_batches;
  return VMC_OK;
}

int vmc_sr_num_stored(vmc_ctx* c, int32_t* n) {
  CHECK_CTX(c);
  if (!n) return fail(c, VMC_ERR_INVALID, "null");
  *n = c->sr_n;
  return VMC_OK;
}

static int sr_read_rr(vmc_ctx* c, int idx, double* rr) {
  if (!rr) return VMC_OK;
  HIPCHK(c, hipMemcpyAsync(rr, c->sr_sc + idx, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return VMC_OK;
}

int vmc_sr_begin(vmc_ctx* c, double* rr0) {
  ENTER(c);
  c->cg_sr_tape_rows = 0;        // a new solve: the general convolution path re-runs the stored chains' taped forward once
  if (c->sr_cap <= 0) return fail(c, VMC_ERR_STATE, "vmc_sr_reserve first");
  if (c->sr_n <= 0) return fail(c, VMC_ERR_STATE, "no samples recorded (vmc_accumulate in ENERGY_GRADIENT mode)");
  PROPAGATE(acc_zeros(c));
  HIPCHK(c, launch_sr_rhs(c->stream, c->acc, (int)c->P, c->sr_x, c->sr_r, c->sr_p, c->sr_partial, c->sr_sc));
  c->sr_iter = 0; c->sr_begun = true;
  return sr_read_rr(c, 0, rr0);
}

// u[0..P) = sum over this rank's stored samples of (O_b . p) O_b,  u[P] = sum (O_b . p)
int vmc_sr_matvec_partial(vmc_ctx* c) {
  ENTER(c);
  if (!c->sr_begun) return fail(c, VMC_ERR_STATE, "vmc_sr_begin first");
  const int B = c->B, N = c->N, H = c->H, Hp = c->Hp, L = c->A;
  const long long R = (long long)c->sr_cap * B;   // row stride between layers of the store
  const int rows = c->sr_n * B;                   // all recorded samples in one pass
  const float* v = c->sr_p;
  Timer t(c, "sr_matvec");
  HIPCHK(c, hipMemsetAsync(c->sr_u, 0, (c->P + 1) * sizeof(float), c->stream));
  if (c->conv_general) {
    if (!c->sr_centre)
      return fail(c, VMC_ERR_UNSUPPORTED, "on the general convolution path the op-by-op matvec is vmc_sr_matvec_phase1 -> all-reduce of the buffer's "
                                          "last float -> vmc_sr_matvec_phase2 (its per-sample weights are centred on the mean over ALL ranks, which "
                                          "vmc_sr_matvec_partial cannot know); or vmc_sr_solve / vmc_sr_solve_dist");
    return cgen_sr_matvec(c, v, rows);
  }
  if (c->conv) {
    // t_b = O_b . p: the CG direction packed like a parameter set, convolved with the taped inputs and
    // dotted with the stored deltas (k_conv_sr_rowdot); u = sum_b t_b O_b: the weight-gradient kernel
    // over the stored samples with per-sample weight t_b (the unweighted sum is skipped)
    const long long Rc = (long long)c->sr_cap * B;
    HIPCHK(c, launch_conv_pack(c->stream, v, c->cg, c->sr_cw0, c->sr_cwf, c->sr_cwb, c->sr_cbias));
    ConvSrRowdotArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.g = c->cg; ra.p = ConvParams{c->sr_cw0, c->sr_cwf, c->sr_cwb, c->sr_cbias};
    ra.configs = c->sr_cfg; ra.tape = c->sr_ctape; ra.tape_stride = Rc * c->cg.CS;
    ra.delta = c->sr_cdelta; ra.delta_stride = Rc * c->cg.CS; ra.t = c->sr_t; ra.n_rows = rows; ra.G = c->cG;
    HIPCHK(c, launch_conv_sr_rowdot(c->stream, ra, c->num_cus));
    ConvDwArgs dw;
    memset(&dw, 0, sizeof(dw));
    dw.g = c->cg; dw.configs = c->sr_cfg; dw.tape = c->sr_ctape; dw.tape_stride = Rc * c->cg.CS;
    dw.delta = c->sr_cdelta; dw.delta_stride = Rc * c->cg.CS; dw.w = c->sr_t; dw.B = rows;
    dw.n_slices = c->sr_cslices < rows ? c->sr_cslices : rows; dw.ws = c->sr_cws; dw.g1 = nullptr; dw.g2 = c->sr_u;
    HIPCHK(c, launch_conv_dw(c->stream, dw));
    HIPCHK(c, launch_sr_tsum(c->stream, c->sr_t, rows, c->sr_u + c->P));
    return VMC_OK;
  }
  // t_b = O_b . p = sum_l delta_l[b] . (a_{l-1}[b] V_l + v_l) + (output / onsite layer term);
  // the row-dot kernel takes <= 256 output units at a time (257 .. 512 units: two column blocks)
  // every (layer, column block) writes its own partial t: ONE launch for all of them (no round of
  // the chip left a quarter full per layer); the output / onsite term folds the partials in the order
  // in which they used to be added into t
  {
    std::vector<SrRowdotArgs> probs;
    for (int l = 0; l < L; ++l) {
      const float* a_in = l == 0 ? c->sr_cfg : c->sr_act + (long long)(l - 1) * R * Hp;
      for (int n0 = 0; n0 < H; n0 += 256) {
        const int nb = H - n0 < 256 ? H - n0 : 256;
        SrRowdotArgs g{a_in, l == 0 ? N : Hp, v + off_w(c, l) + n0, H, v + off_b(c, l) + n0,
                       c->sr_delta + (long long)l * R * Hp + n0, Hp, c->sr_tpart + (long long)probs.size() * R,
                       rows, nb, (int)(l == 0 ? N : H), 1};
        probs.push_back(g);
      }
    }
    HIPCHK(c, launch_sr_rowdot_batch(c->stream, probs.data(), (int)probs.size()));
    const int np = (int)probs.size();
    if (c->rbm)
      HIPCHK(c, launch_sr_row_linear(c->stream, c->sr_cfg, N, v + c->lay.off_won, v + off_bout(c), rows, N, c->sr_t,
                                     c->sr_tpart, np, R));
    else
      HIPCHK(c, launch_sr_row_linear(c->stream, c->sr_act + (long long)(L - 1) * R * Hp, Hp, v + off_wout(c),
                                     v + off_bout(c), rows, H, c->sr_t, c->sr_tpart, np, R));
  }
  // u = sum_b t_b O_b: per layer [a_{l-1} | 1]^T (t (.) delta_l), written in the theta layout, in
  // (<= 256 input rows) x (<= 256 output units) blocks; the bias row comes with the first row block
  const int slices = sr_wsum_slices(rows, c->num_cus);
  for (int l = 0; l < L; ++l) {
    const float* a_in = l == 0 ? c->sr_cfg : c->sr_act + (long long)(l - 1) * R * Hp;
    const int M = l == 0 ? N : H;
    for (int m0 = 0; m0 < M; m0 += 256)
      for (int n0 = 0; n0 < H; n0 += 256) {
        const int mb = M - m0 < 256 ? M - m0 : 256, nb = H - n0 < 256 ? H - n0 : 256;
        HIPCHK(c, launch_sr_wsum(c->stream, a_in + m0, l == 0 ? N : Hp, c->sr_delta + (long long)l * R * Hp + n0, Hp,
                                 c->sr_t, c->sr_ws, c->sr_u + off_w(c, l) + (long long)m0 * H + n0, H,
                                 m0 == 0 ? c->sr_u + off_b(c, l) + n0 : nullptr, mb, nb, rows, slices));
      }
  }
  // the N = 1 layer (w_out, b_out of fully_connected; w_on, b_on of rbm: weights then bias in
  // theta) and sum_b t_b in one column-sum pass
  if (c->rbm)
    HIPCHK(c, launch_sr_colsum(c->stream, c->sr_cfg, N, c->sr_t, rows, N, c->sr_ws, slices,
                               c->sr_u + c->lay.off_won, c->sr_u + c->P));
  else
    HIPCHK(c, launch_sr_colsum(c->stream, c->sr_act + (long long)(L - 1) * R * Hp, Hp, c->sr_t, rows, H,
                               c->sr_ws, slices, c->sr_u + off_wout(c), c->sr_u + c->P));
  return VMC_OK;
}

int vmc_sr_buffer_devptr(vmc_ctx* c, void** dev_ptr, int64_t* n_floats) {
  CHECK_CTX(c);
  if (!c->sr_u) return fail(c, VMC_ERR_STATE, "vmc_sr_reserve first");
  if (dev_ptr) *dev_ptr = c->sr_u;
  if (n_floats) *n_floats = c->P + 1;
  return VMC_OK;
}

int vmc_sr_get_buffer(vmc_ctx* c, float* host) {
  ENTER(c);
  if (!host || !c->sr_u) return fail(c, VMC_ERR_INVALID, "null / vmc_sr_reserve first");
  HIPCHK(c, hipMemcpyAsync(host, c->sr_u, (c->P + 1) * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return VMC_OK;
}

int vmc_sr_set_buffer(vmc_ctx* c, const float* host) {
  ENTER(c);
  if (!host || !c->sr_u) return fail(c, VMC_ERR_INVALID, "null / vmc_sr_reserve first");
  HIPCHK(c, hipMemcpyAsync(c->sr_u, host, (c->P + 1) * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return VMC_OK;
}

int vmc_sr_cg_update(vmc_ctx* c, float diag_shift, double* rr) {
  ENTER(c);
  if (!c->sr_begun) return fail(c, VMC_ERR_STATE, "vmc_sr_begin first");
  const int cur = c->sr_iter & 1;
  HIPCHK(c, launch_sr_q(c->stream, c->sr_u, c->acc, (int)c->P, c->sr_p, diag_shift, c->sr_q, c->sr_partial, c->sr_sc));
  HIPCHK(c, launch_sr_step(c->stream, c->sr_sc, cur, (int)c->P, c->sr_p, c->sr_q, c->sr_x, c->sr_r, c->sr_partial));
  c->sr_iter += 1;
  return sr_read_rr(c, cur ^ 1, rr);
}

// sr_centre for the extent of a solve / debug matvec, whatever way the function leaves (an error return inside the CG
// loop used to leave it set: a later op-by-op vmc_sr_matvec_partial of a sharded caller would then have run the
// single-rank centred matvec instead of being refused)
struct SrCentreScope {
  vmc_ctx* c;
  SrCentreScope(vmc_ctx* ctx, bool on) : c(ctx) { c->sr_centre = on; }
  ~SrCentreScope() { c->sr_centre = false; }
  SrCentreScope(const SrCentreScope&) = delete;
  SrCentreScope& operator=(const SrCentreScope&) = delete;
};

// The op-by-op matvec in two phases (every path; only the general convolution path needs the pair):
//   phase 1  general convolutions: t_b = O_b . p of this rank's stored samples, buffer[P] = sum_b t_b, buffer[0 .. P) = 0;
//            elsewhere nothing
//   -- the caller all-reduces buffer[P] (one float) when the samples are sharded --
//   phase 2  general convolutions: the weights t_b centred on buffer[P] / (samples over all ranks), buffer[0 .. P) =
//            sum_b (t_b - mean) O_b; elsewhere vmc_sr_matvec_partial
// followed, as after vmc_sr_matvec_partial, by the all-reduce of the whole buffer and vmc_sr_cg_update.
int vmc_sr_matvec_phase1(vmc_ctx* c) {
  ENTER(c);
  if (!c->sr_begun) return fail(c, VMC_ERR_STATE, "vmc_sr_begin first");
  if (!c->conv_general) return VMC_OK;
  const int rows = c->sr_n * c->B;
  Timer t(c, "sr_matvec");
  HIPCHK(c, hipMemsetAsync(c->sr_u, 0, (c->P + 1) * sizeof(float), c->stream));
  PROPAGATE(cgen_sr_phase1(c, c->sr_p, rows));
  HIPCHK(c, launch_sr_tsum(c->stream, c->sr_t, rows, c->sr_u + c->P));
  c->sr_phase1_done = true;
  return VMC_OK;
}

int vmc_sr_matvec_phase2(vmc_ctx* c) {
  ENTER(c);
  if (!c->sr_begun) return fail(c, VMC_ERR_STATE, "vmc_sr_begin first");
  if (!c->conv_general) return vmc_sr_matvec_partial(c);
  if (!c->sr_phase1_done) return fail(c, VMC_ERR_STATE, "vmc_sr_matvec_phase1 first");
  c->sr_phase1_done = false;
  const int rows = c->sr_n * c->B;
  Timer t(c, "sr_matvec");
  // (acc[2 P + 1]: the number of samples behind the accumulators -- over all ranks once they are all-reduced, which
  // vmc_sr_begin requires)
  HIPCHK(c, launch_cgen_tcentre_global(c->stream, c->sr_t, rows, c->acc + 2 * c->P + 1, c->cg_centre, c->sr_u + c->P));
  return cgen_sr_phase2(c, rows);
}

static int sr_solve_impl(vmc_ctx* c, void* comm, int world, float diag_shift, float tol, int32_t max_iter,
                         int32_t* iters, double* rel_residual) {
  if (max_iter < 0 || tol < 0.f) return fail(c, VMC_ERR_INVALID, "bad CG arguments");
  double rr0 = 0.0, rr = 0.0;
  PROPAGATE(vmc_sr_begin(c, &rr0));
  rr = rr0;
  int it = 0;
  SrCentreScope centre(c, !sharded(comm, world));   // (general convolution path: see cgen_sr_matvec)
  while (it < max_iter && rr > (double)tol * (double)tol * rr0 && rr0 > 0.0) {
    if (c->conv_general && sharded(comm, world)) {
      // the general convolution path centres its weights on the mean of O_b . p over ALL ranks (cgen_sr_matvec): one more
      // all-reduce, of sum_b O_b . p alone, between its two phases
      const int rows = c->sr_n * c->B;
      HIPCHK(c, hipMemsetAsync(c->sr_u, 0, (c->P + 1) * sizeof(float), c->stream));
      PROPAGATE(cgen_sr_phase1(c, c->sr_p, rows));
      HIPCHK(c, launch_sr_tsum(c->stream, c->sr_t, rows, c->sr_u + c->P));
      PROPAGATE(reduce_buffer(c, comm, world, c->sr_u + c->P, 1, VMC_REDUCE_SUM));
      HIPCHK(c, launch_cgen_tcentre_global(c->stream, c->sr_t, rows, c->acc + 2 * c->P + 1, c->cg_centre, c->sr_u + c->P));
      PROPAGATE(cgen_sr_phase2(c, rows));
    } else
    PROPAGATE(vmc_sr_matvec_partial(c));
    // sharded samples: u = sum_b (O_b . p) O_b and sum_b O_b . p over all ranks, in stream
    if (sharded(comm, world)) PROPAGATE(reduce_buffer(c, comm, world, c->sr_u, c->P + 1, VMC_REDUCE_SUM));
    PROPAGATE(vmc_sr_cg_update(c, diag_shift, &rr));
    ++it;
  }
  if (iters) *iters = it;
  if (rel_residual) *rel_residual = rr0 > 0.0 ? sqrt(rr / rr0) : 0.0;
  return VMC_OK;
}

int vmc_sr_solve(vmc_ctx* c, float diag_shift, float tol, int32_t max_iter, int32_t* iters, double* rel_residual) {
  ENTER(c);
  return sr_solve_impl(c, nullptr, 1, diag_shift, tol, max_iter, iters, rel_residual);
}

int vmc_sr_solve_dist(vmc_ctx* c, void* nccl_comm, int32_t world_size, float diag_shift, float tol,
                      int32_t max_iter, int32_t* iters, double* rel_residual) {
  ENTER(c);
  return sr_solve_impl(c, nccl_comm, world_size, diag_shift, tol, max_iter, iters, rel_residual);
}

int vmc_sr_get_solution(vmc_ctx* c, float* x) {
  ENTER(c);
  if (!x || !c->sr_x) return fail(c, VMC_ERR_INVALID, "null / vmc_sr_reserve first");
  HIPCHK(c, hipMemcpyAsync(x, c->sr_x, c->P * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return VMC_OK;
}

int vmc_sr_apply(vmc_ctx* c, float lr, double* energy) {
  ENTER(c);
  if (!c->sr_begun) return fail(c, VMC_ERR_STATE, "vmc_sr_begin / vmc_sr_solve first");
  HIPCHK(c, launch_sr_apply(c->stream, c->ps[0].theta, c->sr_x, lr, (int)c->P));
  c->ps[0].packed_valid = c->ps[0].cache_valid = false;
  c->acts_valid = false;
  c->sr_begun = false;
  if (energy) PROPAGATE(vmc_mean_energy(c, energy));
  return VMC_OK;
}

int vmc_sr_debug_matvec(vmc_ctx* c, const float* v, float diag_shift, float* out) {
  ENTER(c);
  if (!v || !out) return fail(c, VMC_ERR_INVALID, "null");
  PROPAGATE(vmc_sr_begin(c, nullptr));
  HIPCHK(c, hipMemcpyAsync(c->sr_p, v, c->P * sizeof(float), hipMemcpyHostToDevice, c->stream));
  {
    SrCentreScope centre(c, true);
    PROPAGATE(vmc_sr_matvec_partial(c));
  }
  HIPCHK(c, launch_sr_q(c->stream, c->sr_u, c->acc, (int)c->P, c->sr_p, diag_shift, c->sr_q, c->sr_partial, c->sr_sc));
  HIPCHK(c, hipMemcpyAsync(out, c->sr_q, c->P * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->sr_begun = false;
  return VMC_OK;
}


}  // extern "C"
