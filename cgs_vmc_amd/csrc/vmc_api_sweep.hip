// Sampler entry points of the C ABI (graph_builders.py:38-151): vmc_mc_steps and its injected / debug variants, on the
// fused dense samplers (k_sweep16 / k_sweep8 / the split experiment), the fused convolutional sampler, and the general
// paths (wide.hip, conv_general.hip).  Split out of vmc_api.hip in round 6.
#include "vmc_ctx.hpp"

using namespace vmcapi;

namespace vmcapi {

// One sampler launch: reads the current chain buffers, writes the alternate set, swaps.
//   overtake: the launch goes to sweep_stream and only waits for `dep` (an event on `stream`)
// fc_layer_size > 256: one mc_step = proposals, candidate first layer, H x H GEMMs, output dot,
// accept -- a handful of launches per step, chain state updated in place
// the sampler launch of this ctx (the 3 x bf16 split sampler when it is switched on)
static hipError_t launch_sampler(vmc_ctx* c, hipStream_t st, SweepArgs& a, int which) {
  if (c->split_sweep) { a.p16s = c->ps[which].p16s; return launch_sweep16_split(st, a); }
  // (injected proposals, the proposal dump and the diagnostic stamps stay on k_sweep16: the same chains, bit for bit)
  if (c->sweep_tile == 8 && !a.inj_up && !a.dbg_up && !a.acc_mask) return launch_sweep8(st, a, c->Hp);
  return launch_sweep16(st, a, c->Hp);
}

static int run_sweep_wide(vmc_ctx* c, long long n_steps, bool injected, bool dbg, int* dbg_up, int* dbg_dn,
                          float* dbg_u, unsigned long long step0, bool count_accepted) {
  ParamSet& p = c->ps[0];
  const int B = c->B, N = c->N, H = c->H, Hp = c->Hp, NH = c->n_hh;
  const uint32_t seed_lo = (uint32_t)(c->d.seed & 0xFFFFFFFFull), seed_hi = (uint32_t)(c->d.seed >> 32);
  if (dbg) {
    HIPCHK(c, launch_wide_propose(c->stream, c->configs, B, N, seed_lo, seed_hi, c->d.chain_offset, step0, nullptr,
                                  nullptr, nullptr, dbg_up, dbg_dn, dbg_u));
    return VMC_OK;
  }
  PROPAGATE(ensure_cache(c, VMC_PSI));
  if (count_accepted) HIPCHK(c, hipMemsetAsync(c->d_accepted, 0, sizeof(unsigned long long), c->stream));
  Timer t(c, "sweep");
  // Per step ONE k_wide_step launch (accept the move in flight, propose the next, candidate activations) and
  // the H x H layers as GEMMs; a segment of steps ends with an accept-only launch.
  WideStepArgs w; memset((void*)&w, 0, sizeof(w));
  w.configs = c->configs; w.z1 = p.z1; w.w1p = p.w1p; w.a_last = c->wbuf[NH & 1]; w.a0 = c->wbuf[0];
  w.wout = p.woutp; w.bout = p.bout; w.logit = p.logit;
  w.iup = c->wide_iup; w.idn = c->wide_idn; w.u = c->wide_u;
  if (injected) { w.inj_up = c->inj_up; w.inj_dn = c->inj_dn; w.inj_u = c->inj_u; w.acc_mask = c->acc_mask; }
  w.accepted = c->d_accepted;
  if (c->rbm) { w.onsite = p.onsite; w.won = p.won; }
  w.B = B; w.N = N; w.H = H; w.Hp = Hp; w.act = wide_stage_act(c, 0); w.oact = c->oact;
  w.seed_lo = seed_lo; w.seed_hi = seed_hi; w.chain_offset = c->d.chain_offset;
  bool in_flight = false;                          // a proposal whose last-layer activations are in a_last
  for (long long st = 0; st < n_steps; ++st) {
    if (st > 0 && st % 128 == 0) {   // z1 is updated incrementally: re-derive it from the spins now and then
      w.do_accept = 1; w.do_propose = 0;
      HIPCHK(c, launch_wide_step(c->stream, w));
      in_flight = false;
      p.cache_valid = false;
      PROPAGATE(ensure_cache(c, VMC_PSI));
    }
    w.do_accept = in_flight ? 1 : 0; w.do_propose = 1; w.step = step0 + (unsigned long long)st;
    HIPCHK(c, launch_wide_step(c->stream, w));
    for (int l = 1; l <= NH; ++l) {
      GemmArgs g; memset(&g, 0, sizeof(g));
      g.A = c->wbuf[(l - 1) & 1]; g.sam = Hp; g.sak = 1;
      g.B = p.theta + off_w(c, l); g.sbk = H; g.sbn = 1;
      g.M = B; g.N = H; g.K = H; g.C = c->wbuf[l & 1]; g.ldc = Hp;
      g.bias = p.theta + off_b(c, l); g.epilogue = 1; g.splitk = 1; g.act = wide_stage_act(c, l);
      if (l == NH && wide_rowdot(c, p, g)) { w.dot_part = c->wide_dot; w.n_part = gemm_rowdot_tiles(H); }
      HIPCHK(c, launch_gemm(c->stream, g));
    }
    in_flight = true;
  }
  if (in_flight) {
    w.do_accept = 1; w.do_propose = 0;
    HIPCHK(c, launch_wide_step(c->stream, w));
  }
  c->acts_valid = false;
  c->acc_since_sweep = false;
  return VMC_OK;
}

// z1 / logit (/ onsite) cache of parameter set `which` for the current chains by ONE refresh pass of the sampler
// kernel (n_steps = 0: z1 from the spins, the layers, the output): the 4096 chains of config 3 as 256
// sixteen-chain tiles in one forward (~20 us) where first-layer GEMM + row kernel over 128 units of 32 rows
// take 92 (LogOverlapITSWO's supervisor amplitudes).  The chains are not touched (the kernel's copy of them
// goes to the buffer the next sampler launch overwrites anyway); nothing is swapped.
bool sampler_refresh_ok(const vmc_ctx* c) {
  static const bool on = !(getenv("CGS_VMC_SAMPLER_REFRESH") && atoi(getenv("CGS_VMC_SAMPLER_REFRESH")) == 0);
  return on && !c->conv && !(c->wide && !c->wide_fast);
}
int refresh_cache_by_sampler(vmc_ctx* c, int which) {
  PROPAGATE(ensure_packed(c, which));
  ParamSet& p = c->ps[which];
  SweepArgs a;
  memset(&a, 0, sizeof(a));
  a.pp = p.packed();
  a.configs_in = c->configs; a.z1_in = p.z1; a.logit_in = p.logit;
  a.configs = c->configs_alt; a.z1 = p.z1; a.logit = p.logit;
  a.onsite = p.onsite; a.rbm = c->rbm ? 1 : 0;
  a.accepted = c->d_accepted;
  a.B = c->B; a.N = c->N; a.n_hidden = c->n_hh;
  a.chain_offset = c->d.chain_offset;
  a.seed_lo = (uint32_t)(c->d.seed & 0xFFFFFFFFull); a.seed_hi = (uint32_t)(c->d.seed >> 32);
  a.step0 = c->step; a.n_steps = 0;
  a.waves = c->sweep_waves; a.no_w1l = c->sweep_no_w1l;
  a.act = c->hact; a.oact = c->oact;
  a.cache_in_valid = 0;
  Timer t(c, "refresh");
  HIPCHK(c, launch_sampler(c, c->stream, a, which));
  p.cache_valid = true;
  return VMC_OK;
}

// Does a plain launch of n_steps steps take the patch sampler (conv_patch.hip)?  (cgen_patch_mode: CGS_VMC_CONV_PATCH)
static bool cgen_patch_use(const vmc_ctx* c, long long n_steps) {
  const int mode = cgen_patch_mode(c);
  return n_steps >= 1 && (mode == 2 || (mode == 1 && plan_cgen_patch_pays(c->cg) && n_steps >= 8));
}

// Chain groups of the general convolution sampler (plan.hpp's rule; CGS_VMC_CONV_GENERAL_GROUPS=1..4 forces, read per call)
static int cgen_sweep_groups(const vmc_ctx* c) {
  int G = plan_cgen_sweep_groups(c->cg, c->B, c->num_cus);
  if (const char* e = getenv("CGS_VMC_CONV_GENERAL_GROUPS")) G = atoi(e);
  if (G < 1) G = 1;
  if (G > 4) G = 4;
  if (G > c->B) G = (int)c->B;
  return G;
}

// The sampler of the general convolution path: per mc_step the proposals (k_wide_propose: the Philox streams and the
// arg-max / arg-min rule of every sampler here), a full forward of the B candidates (the exchanged pair negated as the
// first convolution gathers its operand), the Metropolis test and commit.  In place on configs / logit.
static int run_sweep_cgen(vmc_ctx* c, long long n_steps, bool injected, bool dbg, int* dbg_up, int* dbg_dn,
                          float* dbg_u, unsigned long long step0, bool count_accepted) {
  ParamSet& p = c->ps[0];
  const int B = c->B, N = c->N;
  const uint32_t seed_lo = (uint32_t)(c->d.seed & 0xFFFFFFFFull), seed_hi = (uint32_t)(c->d.seed >> 32);
  if (dbg) {
    HIPCHK(c, launch_wide_propose(c->stream, c->configs, B, N, seed_lo, seed_hi, c->d.chain_offset, step0, nullptr,
                                  nullptr, nullptr, dbg_up, dbg_dn, dbg_u));
    return VMC_OK;
  }
  PROPAGATE(ensure_cache(c, VMC_PSI));
  if (count_accepted) HIPCHK(c, hipMemsetAsync(c->d_accepted, 0, sizeof(unsigned long long), c->stream));
  Timer t(c, "sweep");
  // One launch closes a step and opens the next (k_cgen_step_tail: map sum, candidate logit, accept, next proposal) where
  // the candidates are one block of rows whose last map cgen_forward leaves in place; injected proposals (test hook) and
  // CGS_VMC_CONV_STEP_TAIL=0 take the four separate launches
  const char* tail_env = getenv("CGS_VMC_CONV_STEP_TAIL");
  // The patch sampler (conv_patch.hip): on a lattice much larger than the convolutions' reach a step recomputes the two boxes
  // the exchanged pair touches instead of the lattice, all steps of a chain in one launch -- the same chains bit for bit.
  // (cgen_patch_use: CGS_VMC_CONV_PATCH)
  {
    if (!injected && cgen_patch_use(c, n_steps)) {   // (mc_steps always samples psi: ps[0])
      HIPCHK(c, launch_wide_propose(c->stream, c->configs, B, N, seed_lo, seed_hi, c->d.chain_offset, step0, nullptr, nullptr,
                                    nullptr, c->wide_iup, c->wide_idn, c->wide_u));
      PROPAGATE(cgen_patch_maps(c, VMC_PSI));       // the chains' maps as they stand: one taped full forward
      CgenPatchArgs a;
      cgen_patch_args(c, VMC_PSI, &a);
      a.logit = p.logit; a.iup = c->wide_iup; a.idn = c->wide_idn; a.u = c->wide_u;
      a.accepted = c->d_accepted; a.seed_lo = seed_lo; a.seed_hi = seed_hi; a.chain_offset = c->d.chain_offset;
      a.step0 = step0; a.n_steps = n_steps;
      unsigned long long* d_prof = nullptr;         // diagnostic: the phase clocks of chain 0 to stderr (synchronises)
      if (getenv("CGS_VMC_CONV_PATCH_PROF") && atoi(getenv("CGS_VMC_CONV_PATCH_PROF")) != 0) {
        HIPCHK(c, hipMalloc(&d_prof, 6 * sizeof(unsigned long long)));
        a.prof = d_prof;
      }
      HIPCHK(c, launch_cgen_patch_sweep(c->stream, a));
      if (d_prof) {
        unsigned long long h[6];
        HIPCHK(c, hipMemcpyAsync(h, d_prof, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipFree(d_prof);
        fprintf(stderr, "k_cgen_patch_sweep, clocks per step of chain 0: first convolution %.0f, staging %.0f, tiles %.0f, map sum %.0f, "
                "test + commit %.0f, proposal %.0f\n", (double)h[0] / n_steps, (double)h[1] / n_steps, (double)h[2] / n_steps,
                (double)h[3] / n_steps, (double)h[4] / n_steps, (double)h[5] / n_steps);
      }
      c->acts_valid = false;
      c->acc_since_sweep = false;
      return VMC_OK;
    }
  }
  if (!injected && !(tail_env && atoi(tail_env) == 0) && cgen_single_block(c, B)) {
    const ConvGeom& g = c->cg;
    HIPCHK(c, launch_wide_propose(c->stream, c->configs, B, N, seed_lo, seed_hi, c->d.chain_offset, step0, nullptr, nullptr,
                                  nullptr, c->wide_iup, c->wide_idn, c->wide_u));
    // Chain groups: the chains are independent (graph_builders.py:57-88 maps one update over the batch), so the batch is
    // cut into G contiguous groups whose steps run on streams of their own.  A launch of one group that leaves CUs idle
    // (800 row tiles of 128 positions on 256 CUs: the fourth round is one eighth full) then runs beside the next launch
    // of another group, and launches that are latency (a 36 x 36 lattice at 32 chains) hide each other's.  Every row is
    // computed by the same kernel in the same k order wherever its tile lies: the chains do not depend on G
    // (tests/test_gpu_conv_general.py).  How many: plan_cgen_sweep_groups.
    const int G = cgen_sweep_groups(c);
    if (G > 1) {
      for (int i = 0; i < G - 1; ++i)
        if (!c->cg_grp_stream[i]) HIPCHK(c, hipStreamCreateWithFlags(&c->cg_grp_stream[i], hipStreamNonBlocking));
      for (int i = 0; i < G; ++i)
        if (!c->cg_grp_ev[i]) HIPCHK(c, hipEventCreateWithFlags(&c->cg_grp_ev[i], hipEventDisableTiming));
      HIPCHK(c, hipEventRecord(c->cg_grp_ev[0], c->stream));
      for (int i = 0; i < G - 1; ++i) HIPCHK(c, hipStreamWaitEvent(c->cg_grp_stream[i], c->cg_grp_ev[0], 0));
    }
    int rc = VMC_OK;
    for (long long st = 0; st < n_steps && rc == VMC_OK; ++st) {
      for (int gi = 0; gi < G && rc == VMC_OK; ++gi) {
        const long long r0 = (long long)B * gi / G, r1 = (long long)B * (gi + 1) / G;
        const int rows = (int)(r1 - r0);
        hipStream_t s = gi ? c->cg_grp_stream[gi - 1] : c->stream;
        c->cg_stream_cur = s; c->cg_map_row0 = r0;
        rc = cgen_forward(c, VMC_PSI, c->configs, nullptr, rows, c->wide_iup, c->wide_idn, false, nullptr, nullptr, 0, r0);
        const float* last = cgen_last_map(c);
        c->cg_stream_cur = nullptr; c->cg_map_row0 = 0;
        if (rc != VMC_OK) break;
        if (launch_cgen_step_tail(s, last, N, g.F, cgen_fp(g), c->configs + r0 * N, p.logit + r0, rows, c->oact,
                                  c->wide_iup + r0, c->wide_idn + r0, c->wide_u + r0, c->d_accepted, seed_lo, seed_hi,
                                  c->d.chain_offset + (int)r0, step0 + (unsigned long long)st + 1,
                                  st + 1 < n_steps) != hipSuccess)
          rc = fail(c, VMC_ERR_HIP, "k_cgen_step_tail launch");
      }
    }
    // `stream` takes the groups' work back (also behind a failed launch: nothing may be left running on its own)
    for (int i = 1; i < G; ++i) {
      HIPCHK(c, hipEventRecord(c->cg_grp_ev[i], c->cg_grp_stream[i - 1]));
      HIPCHK(c, hipStreamWaitEvent(c->stream, c->cg_grp_ev[i], 0));
    }
    if (rc != VMC_OK) return rc;
    c->acts_valid = false;
    c->acc_since_sweep = false;
    return VMC_OK;
  }
  for (long long st = 0; st < n_steps; ++st) {
    HIPCHK(c, launch_wide_propose(c->stream, c->configs, B, N, seed_lo, seed_hi, c->d.chain_offset,
                                  step0 + (unsigned long long)st, injected ? c->inj_up : nullptr,
                                  injected ? c->inj_dn : nullptr, injected ? c->inj_u : nullptr, c->wide_iup,
                                  c->wide_idn, c->wide_u));
    PROPAGATE(cgen_forward(c, VMC_PSI, c->configs, nullptr, B, c->wide_iup, c->wide_idn, false, c->cg_lnew));
    HIPCHK(c, launch_cgen_accept(c->stream, c->configs, p.logit, c->cg_lnew, c->wide_iup, c->wide_idn, c->wide_u, B, N,
                                 c->oact, c->d_accepted, injected ? c->acc_mask : nullptr));
  }
  c->acts_valid = false;
  c->acc_since_sweep = false;
  return VMC_OK;
}

static int run_sweep(vmc_ctx* c, long long n_steps, bool injected, bool dbg, int* dbg_up, int* dbg_dn,
                     float* dbg_u, unsigned long long step0, bool count_accepted = false,
                     bool overtake = false, hipEvent_t dep = nullptr) {
  PROPAGATE(ensure_packed(c, 0));
  if (!dbg) c->cnt_valid = false;   // the chains change (set again below when this launch leaves their census)
  if (c->wide && !c->wide_fast)
    return run_sweep_wide(c, n_steps, injected, dbg, dbg_up, dbg_dn, dbg_u, step0, count_accepted);
  if (c->conv_general)
    return run_sweep_cgen(c, n_steps, injected, dbg, dbg_up, dbg_dn, dbg_u, step0, count_accepted);
  ParamSet& p = c->ps[0];
  SweepArgs a;
  memset(&a, 0, sizeof(a));
  a.pp = p.packed();
  a.configs_in = c->configs; a.z1_in = p.z1; a.logit_in = p.logit;
  a.configs = c->configs_alt; a.z1 = p.z1_alt; a.logit = p.logit_alt;
  a.onsite = p.onsite_alt; a.rbm = c->rbm ? 1 : 0;
  a.accepted = c->d_accepted;
  if (injected) { a.inj_up = c->inj_up; a.inj_dn = c->inj_dn; a.inj_u = c->inj_u; a.acc_mask = c->acc_mask; }
  if (dbg) { a.dbg_up = dbg_up; a.dbg_dn = dbg_dn; a.dbg_u = dbg_u; }
  a.B = c->B; a.N = c->N; a.n_hidden = c->n_hh;
  a.chain_offset = c->d.chain_offset;
  a.seed_lo = (uint32_t)(c->d.seed & 0xFFFFFFFFull); a.seed_hi = (uint32_t)(c->d.seed >> 32);
  a.step0 = step0; a.n_steps = n_steps;
  a.waves = c->sweep_waves; a.no_w1l = c->sweep_no_w1l;
  a.act = c->hact; a.oact = c->oact;
  // the activations of the final chains are handed to the gradient path only when a gradient
  // accumulate has been seen since the previous launch (equilibration / evaluation sweeps skip
  // the [L][B][Hp] write-back; gradient_sums then recomputes them)
  const bool hand_over = !c->conv && !dbg && (injected || c->acc_since_sweep || c->sr_cap > 0);
  a.act_out = hand_over ? c->act_alt : nullptr;
  a.dact_out = hand_over ? c->dact_alt : nullptr;
  a.cache_in_valid = (!dbg && !injected && p.cache_valid) ? 1 : 0;
  // the census of the chains this launch leaves behind (k_bond_count's job; CGS_VMC_SWEEP_CENSUS=0: a launch of its own)
  static const bool census_on = !(getenv("CGS_VMC_SWEEP_CENSUS") && atoi(getenv("CGS_VMC_SWEEP_CENSUS")) == 0);
  const bool census = census_on && !c->conv && !dbg && !injected && c->n_bonds > 0 && c->bonds && c->cnt_alt;
  if (census) {
    a.bonds = c->bonds; a.quarter_jz = c->quarter_jz; a.n_bonds = c->n_bonds;
    a.cnt_out = c->cnt_alt; a.diag_out = c->diag_alt;
  }
  hipStream_t st = overtake ? c->sweep_stream : c->stream;
  if (overtake) HIPCHK(c, hipStreamWaitEvent(st, dep, 0));
  // the device counter is only zeroed when the caller will read it back
  if (count_accepted) HIPCHK(c, hipMemsetAsync(c->d_accepted, 0, sizeof(unsigned long long), st));
  if (c->conv) {
    ConvSweepArgs s;
    memset(&s, 0, sizeof(s));
    s.g = c->cg; s.p = conv_params(p);
    s.configs_in = c->configs; s.logit_in = p.logit; s.configs = c->configs_alt; s.logit = p.logit_alt;
    s.accepted = c->d_accepted;
    s.inj_up = a.inj_up; s.inj_dn = a.inj_dn; s.inj_u = a.inj_u; s.acc_mask = a.acc_mask;
    s.dbg_up = a.dbg_up; s.dbg_dn = a.dbg_dn; s.dbg_u = a.dbg_u;
    s.oact = c->oact; s.cache_in_valid = a.cache_in_valid; s.B = c->B; s.G = c->cGs;
    s.chain_offset = a.chain_offset; s.seed_lo = a.seed_lo; s.seed_hi = a.seed_hi;
    s.step0 = step0; s.n_steps = n_steps;
    Timer t(c, "sweep", st, true);
    HIPCHK(c, launch_conv_sweep(st, s));
  } else {
    Timer t(c, "sweep", st, true);
    HIPCHK(c, launch_sampler(c, st, a, 0));
  }
  if (dbg) return VMC_OK;               // the proposal dump writes nothing back
  swap_chain_buffers(c);
  c->cnt_valid = census;
  c->acts_valid = hand_over;
  c->acc_since_sweep = false;
  if (overtake) {
    HIPCHK(c, hipEventRecord(c->ev_sweep_done, st));
    c->sweep_pending = true;
  }
  return VMC_OK;
}

}  // namespace vmcapi

extern "C" {

int vmc_debug_conv_patch(vmc_ctx* c, int64_t n_steps, int32_t* patch) {
  CHECK_CTX(c);
  if (!patch) return fail(c, VMC_ERR_INVALID, "null");
  *patch = cgen_patch_use(c, n_steps) ? 1 : 0;
  return VMC_OK;
}

int vmc_mc_steps(vmc_ctx* c, int64_t n_steps, int64_t* accepted) {
  CHECK_CTX(c);
  if (n_steps < 0) return fail(c, VMC_ERR_INVALID, "n_steps < 0");
  if (n_steps == 0) {               // `for _ in range(0)`: nothing runs, nothing is launched
    c->side_sweep_once = false;     // (a request for the side stream does not outlive the call it was made for)
    if (accepted) *accepted = 0;
    return VMC_OK;
  }
  // training.py:614-617: accumulate_gradients and the following mc_steps only share the chains
  // R_t, which the sampler reads and never writes in place, so the launch need not wait for the
  // accumulate that was enqueued just before it: it waits for the event recorded when that
  // accumulate STARTED.  Anything else in between (or a re-pack of the parameters) makes it wait
  // for everything enqueued so far.
  const bool after_acc = c->token && c->ps[0].packed_valid;
  c->token = false;
  // side: a sampler that fills the chip (config 3) cannot overtake its accumulate, but it can leave `stream`
  // free for the collective that follows it (epoch_energy_gradient_impl): same launch, other stream, behind
  // an event recorded after everything enqueued so far
  const bool side = c->side_sweep_once && c->overlap && !can_overlap(c);
  c->side_sweep_once = false;
  const bool overtake = can_overlap(c) || side;
  hipEvent_t dep = c->ev_mark;
  if (overtake && (!after_acc || side)) {
    PROPAGATE(ensure_packed(c, 0));
    HIPCHK(c, hipEventRecord(c->ev_now, c->stream));
    dep = c->ev_now;
  }
  if (!overtake) PROPAGATE(join_sweep(c));
  c->expect_sweep = overtake && after_acc && !side;
  PROPAGATE(run_sweep(c, n_steps, false, false, nullptr, nullptr, nullptr, c->step, accepted != nullptr,
                      overtake, dep));
  c->step += (unsigned long long)n_steps;
  c->ps[0].cache_valid = !c->wide || c->wide_fast;   // the sweep kernel writes back an exact z1/logit cache (the general
                                     // wide path keeps an incrementally updated one: recomputed on demand)
  c->ps[1].cache_valid = false;
  c->list_valid = false;
  if (accepted) {
    hipStream_t st = overtake ? c->sweep_stream : c->stream;
    unsigned long long h = 0;
    HIPCHK(c, hipMemcpyAsync(&h, c->d_accepted, sizeof(h), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    *accepted = (int64_t)h;
  }
  return VMC_OK;
}

int vmc_mc_step_injected(vmc_ctx* c, const int32_t* i_up, const int32_t* i_dn, const float* u, uint8_t* accept_mask) {
  ENTER(c);
  if (!i_up || !i_dn || !u) return fail(c, VMC_ERR_INVALID, "null proposals");
  for (int b = 0; b < c->B; ++b)
    if (i_up[b] < 0 || i_up[b] >= c->N || i_dn[b] < 0 || i_dn[b] >= c->N)
      return fail(c, VMC_ERR_INVALID, "proposal site out of range");
  HIPCHK(c, hipMemcpyAsync(c->inj_up, i_up, c->B * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->inj_dn, i_dn, c->B * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->inj_u, u, c->B * sizeof(float), hipMemcpyHostToDevice, c->stream));
  PROPAGATE(run_sweep(c, 1, true, false, nullptr, nullptr, nullptr, c->step));
  c->ps[0].cache_valid = !c->wide || c->wide_fast; c->ps[1].cache_valid = false; c->list_valid = false;
  c->cnt_valid = false;
  if (accept_mask)
    HIPCHK(c, hipMemcpyAsync(accept_mask, c->acc_mask, c->B, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return VMC_OK;
}

int vmc_debug_proposals(vmc_ctx* c, uint64_t step, int32_t* i_up, int32_t* i_dn, float* u) {
  ENTER(c);
  if (!i_up || !i_dn || !u) return fail(c, VMC_ERR_INVALID, "null outputs");
  PROPAGATE(run_sweep(c, 0, false, true, c->inj_up, c->inj_dn, c->inj_u, step));
  HIPCHK(c, hipMemcpyAsync(i_up, c->inj_up, c->B * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(i_dn, c->inj_dn, c->B * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(u, c->inj_u, c->B * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return VMC_OK;
}

int vmc_debug_sweep_profile(vmc_ctx* c, int64_t n_steps, double* phase_cycles) {
  ENTER(c);
  if (n_steps < 1 || !phase_cycles) return fail(c, VMC_ERR_INVALID, "bad arguments");
  if (c->rbm || c->conv || c->wide) return fail(c, VMC_ERR_UNSUPPORTED, "the diagnostic sweep build exists for fully_connected (<= 256 units) only");
  PROPAGATE(ensure_packed(c, 0));
  const bool tile8 = c->sweep_tile == 8;      // k_sweep8's stamped instantiation (phases: sweep8.hip)
  const int wpg = tile8 ? c->Hp / 32 : c->sweep_waves;
  const int grid = tile8 ? (c->B + 7) / 8 : (c->B + 15) / 16;
  unsigned long long* d = nullptr;
  HIPCHK(c, dalloc(&d, (long long)grid * 128));
  HIPCHK(c, hipMemsetAsync(d, 0, (size_t)grid * 128 * sizeof(unsigned long long), c->stream));
  SweepArgs a;
  memset(&a, 0, sizeof(a));
  a.pp = c->ps[0].packed();
  a.configs_in = c->configs; a.z1_in = c->ps[0].z1; a.logit_in = c->ps[0].logit;
  a.configs = c->configs_alt; a.z1 = c->ps[0].z1_alt; a.logit = c->ps[0].logit_alt;
  a.accepted = c->d_accepted; a.dbg_cycles = d; a.waves = 8; a.act = c->hact; a.oact = c->oact;
  a.B = c->B; a.N = c->N; a.n_hidden = c->n_hh; a.chain_offset = c->d.chain_offset;
  a.seed_lo = (uint32_t)(c->d.seed & 0xFFFFFFFFull); a.seed_hi = (uint32_t)(c->d.seed >> 32);
  a.step0 = c->step; a.n_steps = n_steps;
  if (tile8) HIPCHK(c, launch_sweep8(c->stream, a, c->Hp));
  else HIPCHK(c, launch_sweep16(c->stream, a, c->Hp));
  swap_chain_buffers(c);
  c->acts_valid = false;
  c->step += (unsigned long long)n_steps;
  c->ps[0].cache_valid = true; c->ps[1].cache_valid = false; c->list_valid = false;
  c->cnt_valid = false;
  std::vector<unsigned long long> h((size_t)grid * 128);
  HIPCHK(c, hipMemcpyAsync(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  hipFree(d);
  // CGS_VMC_PROFILE_WAVES = bit mask of the waves of a workgroup to average over (diagnostic;
  // default all): waves 0-3 own the chains (proposals, accept, Philox), waves 4-7 do not
  unsigned wave_mask = ~0u;
  if (const char* e = getenv("CGS_VMC_PROFILE_WAVES")) wave_mask = (unsigned)strtoul(e, nullptr, 0);
  for (int k = 0; k < 16; ++k) {
    double s = 0.0;
    long long cnt = 0;
    for (int i = 0; i < grid * wpg; ++i)
      if ((wave_mask >> (i % wpg)) & 1u) { s += (double)h[(size_t)i * 16 + k]; ++cnt; }
    phase_cycles[k] = cnt ? s / ((double)cnt * (double)n_steps) : 0.0;
  }
  return VMC_OK;
}

int vmc_get_step_counter(vmc_ctx* c, uint64_t* step) { CHECK_CTX(c); if (!step) return fail(c, VMC_ERR_INVALID, "null"); *step = c->step; return VMC_OK; }
int vmc_set_step_counter(vmc_ctx* c, uint64_t step) { CHECK_CTX(c); c->step = step; return VMC_OK; }


}  // extern "C"
