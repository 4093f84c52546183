// Host side of libcgsvmc_hip.so: the C ABI of include/cgsvmc.h on top of the gfx950 kernels.  One vmc_ctx per GPU, all
// work on ctx->stream.  This file: ctx life cycle, parameters, chains, amplitudes, local energies, timing; the other
// entry points live in vmc_api_sweep.hip (samplers), vmc_api_train.hip (accumulators, Adam, epochs, evaluation),
// vmc_api_coll.hip (collectives), vmc_api_sr.hip (stochastic reconfiguration); vmc_api_cgen.hip is the general
// convolution path's machinery.  Shared state and helpers: vmc_ctx.hpp.
#include "vmc_ctx.hpp"

using namespace vmcapi;

namespace vmcapi { std::string g_create_error; }

namespace vmcapi {

int ensure_packed(vmc_ctx* c, int which) {
  ParamSet& p = c->ps[which];
  if (!p.has_params) return fail(c, VMC_ERR_STATE, "parameters not set (vmc_set_params)");
  if (p.packed_valid) return VMC_OK;
  if (c->conv) {
    // (general path: the parameter slices are the B matrices of its GEMMs as they lie in theta)
    if (!c->conv_general) HIPCHK(c, launch_conv_pack(c->stream, p.theta, c->cg, p.cw0, p.cwf, p.cwb, p.cbias));
    p.packed_valid = true;
    return VMC_OK;
  }
  HIPCHK(c, launch_pack(c->stream, p.theta, c->N, c->H, c->Hp, c->lay, p.w1p, p.b1p, p.bh, p.p16,
                        p.p16t, p.woutp, p.bout, p.won));
  if (c->split) HIPCHK(c, launch_pack_split(c->stream, p.theta, c->H, c->lay, p.p16s));
  p.packed_valid = true;
  return VMC_OK;
}

// rows through the fused row kernel of this ctx (the 3 x bf16 split experiment when it is switched on)
hipError_t launch_rows(vmc_ctx* c, int which, const TailArgs& a, bool ratio) {
  if (c->split) return launch_tail16_split(c->stream, a, c->ps[which].p16s, ratio);
  return launch_tail(c->stream, a, c->Hp, ratio, c->rbm);
}

TailArgs tail_args(vmc_ctx* c, int which) {
  TailArgs a;
  memset(&a, 0, sizeof(a));
  a.pp = c->ps[which].packed();
  a.bonds = c->bonds ? c->bonds : c->bond_dummy;   // the kernel loads bonds[0] unconditionally
  a.half_jx = c->half_jx;
  a.n_hidden = c->n_hh;
  a.n_sites = c->N;
  a.n_units = c->H;
  a.on_base = c->ps[which].onsite;
  a.num_cus = c->num_cus;
  a.act = c->hact; a.oact = c->oact;
  return a;
}

ConvParams conv_params(const ParamSet& p) { return ConvParams{p.cw0, p.cwf, p.cwb, p.cbias}; }

// Conv2DNetwork / ResNet2D forward (wavefunctions.py:596-598, 790-792) of parameter set `which` on
// the rows of a row list over `configs`: logits (ratio == false) or 0.5 jx psi'/psi of the
// bond-exchanged configurations.  with_tape: the inputs of every convolution go to c->ctape.
// The same on the general path (conv_general.hip): blocks of cg_rows row configurations; per convolution an im2col
// gather (explicit, or inside the product's A operand; what the maps hold: cgen_post below) and one
// GEMM against the parameter slice in theta; ResBlock2d's `v + h` (layers.py:227) is the accumulate epilogue.
// iup / idn != nullptr: row r is chain r with that pair exchanged (the sampler's candidates).
int conv_rows(vmc_ctx* c, int which, const float* configs, const int2* rowinfo, int rows,
              const int* rows_dev, bool ratio, float* out, bool with_tape) {
  if (c->conv_general) {
    // (the gradient path of the general convolution re-runs its own taped forward, cgen_gradient: the row entry keeps none)
    if (with_tape) return fail(c, VMC_ERR_UNSUPPORTED, "conv_rows keeps no tape on the general convolution path (its gradient entries run their own taped forward)");
    int n_rows = rows;
    if (rows_dev) {            // the blocks need the row count on the host
      HIPCHK(c, hipMemcpyAsync(&n_rows, rows_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return cgen_forward(c, which, configs, rowinfo, n_rows, nullptr, nullptr, ratio, out);
  }
  ConvRowsArgs a;
  memset(&a, 0, sizeof(a));
  a.g = c->cg; a.p = conv_params(c->ps[which]);
  a.configs = configs; a.rowinfo = rowinfo;
  a.bonds = c->bonds ? c->bonds : c->bond_dummy; a.half_jx = c->half_jx;
  a.logit_base = c->ps[which].logit;
  a.n_rows_dev = rows_dev; a.n_rows = rows; a.ratio = ratio ? 1 : 0; a.oact = c->oact;
  a.G = c->cG; a.out = out;
  a.tape = with_tape ? c->ctape : nullptr; a.tape_stride = c->ctape_stride;
  HIPCHK(c, launch_conv_rows(c->stream, a, c->num_cus));
  return VMC_OK;
}

// First layer of raw configurations on the matrix cores: z1[rows,Hp] = X[rows,N] W1p[N,Hp] + b1
// through the LDS-tiled fp32-MFMA GEMM (64x64x32 tiles of spins and weights staged in LDS,
// dwordx4 loads of the configuration batch).  wavefunctions.py:345-349, first snt.Linear.
int first_layer(vmc_ctx* c, const ParamSet& p, const float* configs, float* z1, int rows) {
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.A = configs; g.sam = c->N; g.sak = 1;
  g.B = p.w1p; g.sbk = c->Hp; g.sbn = 1;
  g.M = rows; g.N = c->Hp; g.K = c->N; g.C = z1; g.ldc = c->Hp;
  g.bias = p.b1p; g.epilogue = 4; g.splitk = 1;
  HIPCHK(c, launch_gemm(c->stream, g));
  return VMC_OK;
}

// activation behind linear stage l (0 = the N x H layer) on the general path
int wide_stage_act(const vmc_ctx* c, int l) {
  return (c->rbm && l == c->n_hh) ? VMC_ACT_LOGCOSH_ : c->hact;
}

// The last H x H layer of a forward whose activations nobody reads: turn its GEMM into the row-dot form (the output
// layer's dot product as column-tile partials in c->wide_dot, nothing stored to C) where a tile kernel takes the
// shape.  CGS_VMC_ROWDOT=0: never (A/B measurements, tests).  Returns whether `g` was changed.
bool wide_rowdot(vmc_ctx* c, const ParamSet& p, GemmArgs& g) {
  const char* e = getenv("CGS_VMC_ROWDOT");
  if (e && atoi(e) == 0) return false;
  GemmArgs t = g;
  t.epilogue = 10; t.dot_w = p.woutp; t.dot_out = c->wide_dot; t.C = nullptr;
  if (!gemm_rowdot_ok(t)) return false;
  g = t;
  return true;
}

// fc_layer_size > 256: rows {chain, bond} of a row list over the cached z1 -> logits / ratios,
// `wrows` rows at a time: rank-2 first layer written out, H x H layers as GEMMs, output dot
// RBM (wavefunctions.py:418-437): the last linear stage goes through log cosh instead of the hidden
// activation, the output "dot" is against ones (+ b_on) and `onsite` holds x . w_on of the base rows
int wide_forward(vmc_ctx* c, int which, const float* z1, const int2* rowinfo, long long n_rows, bool ratio,
                 float* out, const float* onsite) {
  ParamSet& p = c->ps[which];
  const int H = c->H, Hp = c->Hp, NH = c->n_hh;
  const int2* bonds = c->bonds ? c->bonds : c->bond_dummy;
  const WideOnsite on{c->rbm ? onsite : nullptr, p.won, bonds, nullptr, nullptr};
  if (c->wide_fast) {
    TailArgs a = tail_args(c, which);
    a.z1 = z1; a.logit_base = p.logit; a.rowinfo = rowinfo; a.on_base = onsite;
    a.n_rows = (int)n_rows; a.out = out;
    if (NH == 0) HIPCHK(c, launch_tail(c->stream, a, Hp, ratio, c->rbm));      // k_tail0 takes any Hp
    else HIPCHK(c, launch_tail_lds(c->stream, a, Hp, ratio, c->rbm));
    return VMC_OK;
  }
  for (long long row0 = 0; row0 < n_rows; row0 += c->wrows) {
    const int rows = (int)(n_rows - row0 < c->wrows ? n_rows - row0 : c->wrows);
    HIPCHK(c, launch_wide_rows_act(c->stream, z1, p.w1p, rowinfo, bonds, row0, rows,
                                   Hp, wide_stage_act(c, 0), c->wbuf[0]));
    bool dot_fused = false;          // the output dot rode in the last layer's GEMM (row-dot epilogue)
    for (int l = 1; l <= NH; ++l) {
      GemmArgs g; memset(&g, 0, sizeof(g));
      g.A = c->wbuf[(l - 1) & 1]; g.sam = Hp; g.sak = 1;
      g.B = p.theta + off_w(c, l); g.sbk = H; g.sbn = 1;
      g.M = rows; g.N = H; g.K = H; g.C = c->wbuf[l & 1]; g.ldc = Hp;
      g.bias = p.theta + off_b(c, l); g.epilogue = 1; g.splitk = 1; g.act = wide_stage_act(c, l);
      if (l == NH) dot_fused = wide_rowdot(c, p, g);
      HIPCHK(c, launch_gemm(c->stream, g));
    }
    if (dot_fused)
      HIPCHK(c, launch_wide_out_part(c->stream, c->wide_dot, gemm_rowdot_tiles(H), p.bout, rows, rowinfo, row0,
                                     c->half_jx, p.logit, c->oact, ratio, out, on));
    else
    HIPCHK(c, launch_wide_out(c->stream, c->wbuf[NH & 1], p.woutp, p.bout, rows, H, Hp, rowinfo, row0, c->half_jx,
                              p.logit, c->oact, ratio, out, on));
  }
  return VMC_OK;
}

// z1 / logit cache of parameter set `which` for the ctx's chains
int ensure_cache(vmc_ctx* c, int which) {
  PROPAGATE(ensure_packed(c, which));
  ParamSet& p = c->ps[which];
  if (p.cache_valid) return VMC_OK;
  if (c->conv) {
    Timer t(c, "tail_amp");
    const bool tape = which == VMC_PSI && !c->conv_general;       // (the general path keeps no gradient tape)
    PROPAGATE(conv_rows(c, which, c->configs, c->rowinfo_id, c->B, nullptr, false, p.logit, tape));
    if (tape) c->acts_valid = true;
    p.cache_valid = true;
    return VMC_OK;
  }
  {
    Timer t(c, "z1");
    PROPAGATE(first_layer(c, p, c->configs, p.z1, c->B));
    if (c->rbm) HIPCHK(c, launch_onsite(c->stream, c->configs, p.won, c->B, c->N, p.onsite));
  }
  if (c->wide) {
    Timer t(c, "tail_amp");
    PROPAGATE(wide_forward(c, which, p.z1, c->rowinfo_id, c->B, false, p.logit, p.onsite));
  } else {
    Timer t(c, "tail_amp");
    TailArgs a = tail_args(c, which);
    a.z1 = p.z1; a.n_rows = c->B; a.out = p.logit; a.rowinfo = c->rowinfo_id;
    HIPCHK(c, launch_rows(c, which, a, false));
  }
  p.cache_valid = true;
  return VMC_OK;
}

void invalidate_configs(vmc_ctx* c) {
  c->acts_valid = false;
  c->ps[0].cache_valid = c->ps[1].cache_valid = false;
  c->list_valid = false;
  c->cnt_valid = false;
}

int ensure_list(vmc_ctx* c) {
  if (c->n_bonds <= 0) return fail(c, VMC_ERR_STATE, "bonds not set (vmc_set_bonds)");
  if (c->list_valid) return VMC_OK;
  Timer t(c, "bond_list");
  HIPCHK(c, launch_bond_list(c->stream, c->configs, c->bonds, c->quarter_jz, c->B, c->N,
                             c->n_bonds, c->cnt, c->off, c->diag, c->rowinfo, c->cnt_valid));
  c->list_valid = true;
  return VMC_OK;
}

// eloc[which] on device
// defer_reduce: the caller folds the rows into eloc itself (vmc_accumulate lets k_backprop16 do it: one
// dependent launch less per step); *deferred tells whether that is still owed
int local_energy_device(vmc_ctx* c, int which, bool defer_reduce, bool* deferred) {
  if (deferred) *deferred = false;
  PROPAGATE(ensure_cache(c, which));
  PROPAGATE(ensure_list(c));
  ParamSet& p = c->ps[which];
  if (c->conv) {
    Timer t(c, "tail_eloc");
    PROPAGATE(conv_rows(c, which, c->configs, c->rowinfo, (int)((long long)c->B * c->n_bonds), c->off + c->B,
                        true, c->val, false));
  } else if (c->wide && c->wide_fast) {
    Timer t(c, "tail_eloc");
    TailArgs a = tail_args(c, which);
    a.z1 = p.z1; a.logit_base = p.logit; a.rowinfo = c->rowinfo;
    a.n_rows_dev = c->off + c->B;                 // the row count stays on the device
    a.n_rows = (int)((long long)c->B * c->n_bonds);
    a.out = c->val;
    if (c->n_hh == 0) HIPCHK(c, launch_tail(c->stream, a, c->Hp, true, c->rbm));
    else HIPCHK(c, launch_tail_lds(c->stream, a, c->Hp, true, c->rbm));
  } else if (c->wide) {
    Timer t(c, "tail_eloc");
    int n_rows = 0;      // the GEMM grids need the row count on the host
    HIPCHK(c, hipMemcpyAsync(&n_rows, c->off + c->B, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    PROPAGATE(wide_forward(c, which, p.z1, c->rowinfo, n_rows, true, c->val, p.onsite));
  } else {
    Timer t(c, "tail_eloc");
    TailArgs a = tail_args(c, which);
    a.z1 = p.z1; a.logit_base = p.logit; a.rowinfo = c->rowinfo;
    a.n_rows_dev = c->off + c->B;
    a.n_rows = (int)((long long)c->B * c->n_bonds);
    a.out = c->val;
    // a sampler launch is expected to overtake this accumulate: its workgroups need a whole
    // CU each, so the persistent grid leaves them free
    if (c->expect_sweep && can_overlap(c) && c->num_cus - sweep_cus(c) >= c->num_cus / 4) {
      a.num_cus = c->num_cus - sweep_cus(c);
      // Long row lists (config 5: 4,100 tiles of ~170 us) as SHORT-LIVED workgroups instead: a grid of
      // tiles / K workgroups of K tiles each (about CGS_VMC_TAIL_CHUNK_US of work), which the dispatcher hands to
      // whichever CU is free -- the CUs of the sampler too once it has finished (its launch, on the high-priority
      // sweep_stream, takes the CUs the first workgroups free).  A persistent grid on the CUs the sampler leaves
      // cannot grow when the sampler ends before it (eight-chain tiles: 3.6 ms against 5.1 ms on 128 CUs).
      static const int chunk_us = getenv("CGS_VMC_TAIL_CHUNK_US") ? atoi(getenv("CGS_VMC_TAIL_CHUNK_US")) : 150;
      if (chunk_us > 0 && !c->split && c->n_hh > 0) {
        const long long tiles = ((long long)a.n_rows + 127) / 128;
        const double tile_us = 128.0 * c->n_hh * 2.0 * c->Hp * c->Hp / 491520.0;   // 0.8 of a CU's 256 flops per clock at 2.4 GHz
        long long k = (long long)(chunk_us / tile_us);
        if (k < 1) k = 1;
        const long long wgs = (tiles + k - 1) / k;
        if (wgs >= 4LL * c->num_cus) a.num_cus = (int)wgs;
      }
    }
    HIPCHK(c, launch_rows(c, which, a, true));
  }
  if (defer_reduce && deferred && !c->conv && !(c->wide && !c->wide_fast)) {
    *deferred = true;              // the fused back-propagation launch folds val into eloc
    return VMC_OK;
  }
  {
    Timer t(c, "eloc_reduce");
    HIPCHK(c, launch_eloc_reduce(c->stream, c->off, c->diag, c->val, c->B, c->offdiag, p.eloc));
  }
  return VMC_OK;
}

int grow_tmp(vmc_ctx* c, long long rows) {
  if (rows <= c->tmp_rows) return VMC_OK;
  if (c->tmp_cfg) { hipFree(c->tmp_cfg); hipFree(c->tmp_z1); hipFree(c->tmp_out); hipFree(c->tmp_rowinfo); hipFree(c->tmp_on); }
  HIPCHK(c, dalloc(&c->tmp_on, rows));
  HIPCHK(c, dalloc(&c->tmp_rowinfo, rows));
  HIPCHK(c, launch_iota_rows(c->stream, c->tmp_rowinfo, (int)rows));
  HIPCHK(c, dalloc(&c->tmp_cfg, rows * c->N));
  HIPCHK(c, dalloc(&c->tmp_z1, rows * c->Hp));
  HIPCHK(c, dalloc(&c->tmp_out, rows));
  c->tmp_rows = rows;
  return VMC_OK;
}

// layers.NONLINEARITIES on the host (the psi values vmc_amplitude hands back)
float host_activation(int act, float x) {
  switch (act) {
    case VMC_ACT_RELU_: return x > 0.f ? x : 0.f;
    case VMC_ACT_EXP_: return expf(x);
    case VMC_ACT_COS_: return cosf(x);
    case VMC_ACT_TAN_: return tanf(x);
    case VMC_ACT_TANH_: return tanhf(x);
    case VMC_ACT_SIGMOID_: return 1.f / (1.f + expf(-x));
    default: return x;
  }
}


}  // namespace vmcapi

extern "C" {

int64_t vmc_num_params(int32_t n_sites, int32_t layer_size, int32_t num_layers) {
  return plan_num_params_dense(VMC_ANSATZ_FULLY_CONNECTED, n_sites, layer_size, num_layers);
}

int64_t vmc_num_params_ansatz(int32_t ansatz, int32_t n_sites, int32_t layer_size, int32_t num_layers) {
  return plan_num_params_dense(ansatz, n_sites, layer_size, num_layers);
}

int64_t vmc_num_params_conv(int32_t ansatz, int32_t num_layers, int32_t num_filters, int32_t kernel_size) {
  const bool resnet = ansatz == VMC_ANSATZ_RES_NET_2D || ansatz == VMC_ANSATZ_RES_NET_1D;
  const bool one_d = ansatz == VMC_ANSATZ_CONV_1D || ansatz == VMC_ANSATZ_RES_NET_1D;
  const int n_conv = resnet ? 1 + 2 * num_layers : num_layers;
  return conv_num_params(n_conv, num_filters, one_d ? kernel_size : kernel_size * kernel_size);
}

const char* vmc_last_error(const vmc_ctx* ctx) {
  return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int vmc_create(const vmc_desc* d, vmc_ctx** out) {
  if (!d || !out) return fail(nullptr, VMC_ERR_INVALID, "null argument");
  *out = nullptr;
  // every shape decision (ansatz, limits, LDS budgets, padded sizes, parameter layout): plan_desc, plan.hpp
  DescPlan dp;
  {
    char msg[256];
    const char* wf = getenv("CGS_VMC_WIDE_FAST");
    const char* fg = getenv("CGS_VMC_CONV_GENERAL");      // =1: the general convolution path for every shape (tests, A/B runs); =0: only where the fused kernels refuse
    const int rc = plan_desc(d, !(wf && atoi(wf) == 0), &dp, msg, sizeof(msg), fg ? (atoi(fg) != 0 ? 1 : -1) : 0);
    if (rc != VMC_OK) return fail(nullptr, rc, msg);
  }
  const bool rbm = dp.rbm != 0, conv = dp.conv != 0, wide = dp.wide != 0, wide_fast_ok = dp.wide_fast != 0;
  const ConvGeom cg = dp.cg;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(nullptr, VMC_ERR_HIP, "no HIP device available: the VMC hot path has no CPU fallback");
  if (d->device < 0 || d->device >= ndev) return fail(nullptr, VMC_ERR_INVALID, "bad device ordinal");
  if ((e = hipSetDevice(d->device)) != hipSuccess)
    return fail(nullptr, VMC_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));

  vmc_ctx* c = new vmc_ctx();
  c->d = *d;
  c->N = d->n_sites; c->B = d->batch_size; c->L = d->num_layers; c->H = d->layer_size;
  c->rbm = rbm;
  c->conv = conv; c->cg = cg; c->conv_general = conv && dp.conv_general != 0;
  if (conv) { c->L = 1; c->overlap = false; }   // minimal dense-side shapes (unused)
  c->wide = wide;
  if (wide) c->overlap = false;
  c->hact = d->nonlinearity; c->oact = d->output_activation;
  c->lay = dp.lay;
  // 257 .. 512 units (wide_fast): the fused sampler padded to 384 / 512 units (k_sweep16<24|32>), rows on
  // the LDS-operand kernel (k_tail_lds; without an H x H layer: k_tail0) and the fused back-propagation
  // (k_backprop16<24|32>); both dense ansatz types, every hidden activation
  c->wide_fast = wide_fast_ok;
  c->Hp = dp.Hp;
  c->n_hh = c->lay.n_hh; c->A = c->n_hh + 1;
  c->P = dp.P;
  c->stream = (hipStream_t)d->stream;
  if (const char* e = getenv("CGS_VMC_SWEEP_W1L")) c->sweep_no_w1l = atoi(e) == 0 ? 1 : 0;
  if (const char* e = getenv("CGS_VMC_SPLIT_BF16")) {
    c->split = (atoi(e) == 1 || atoi(e) == 2) && !conv && !wide && !rbm && c->Hp == 256 && c->n_hh >= 1 && c->hact == VMC_ACT_RELU_;
    c->split_sweep = c->split && atoi(e) == 2 && sweep16_split_supported(c->N, c->Hp, c->n_hh);
  }
  if (const char* e = getenv("CGS_VMC_OVERLAP")) { c->overlap = !conv && !wide && atoi(e) != 0; c->overlap_full = atoi(e) == 2; }
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d->device) == hipSuccess && prop.multiProcessorCount > 0)
      c->num_cus = prop.multiProcessorCount;
  }
  // eight-chain sampler tiles where sixteen-chain tiles would leave half of the chip idle (sweep8.hip)
  c->sweep8_ok = !conv && !wide && !rbm && c->hact == VMC_ACT_RELU_ && !c->split_sweep &&
                 plan_sweep8(c->N, c->Hp, c->n_hh, c->sweep_no_w1l != 0).ok;
  {
    int forced = 0;
    if (const char* e = getenv("CGS_VMC_SWEEP_TILE")) forced = atoi(e);
    c->sweep_tile = plan_sweep_tile(c->B, c->num_cus, c->sweep8_ok, forced);
  }
  const long long B = c->B, N = c->N, Hp = c->Hp, P = c->P, L = c->A, NH = c->n_hh;   // L: activation buffers
#define CA(expr) do { hipError_t e2 = (expr); if (e2 != hipSuccess) { \
    g_create_error = std::string(#expr) + ": " + hipGetErrorString(e2); vmc_destroy(c); return VMC_ERR_HIP; } } while (0)
  for (int w = 0; w < 2; ++w) {
    ParamSet& p = c->ps[w];
    CA(dalloc(&p.theta, P));
    CA(dalloc(&p.w1p, N * Hp)); CA(dalloc(&p.b1p, Hp)); CA(dalloc(&p.bh, NH * Hp));
    CA(dalloc(&p.p16, (NH > 0 ? NH : 1) * Hp * Hp));
    CA(dalloc(&p.p16t, (NH > 0 ? NH : 1) * Hp * Hp));
    CA(hipMemsetAsync(p.p16t, 0, (size_t)(NH > 0 ? NH : 1) * Hp * Hp * sizeof(float), c->stream));
    CA(hipMemsetAsync(p.p16, 0, (size_t)(NH > 0 ? NH : 1) * Hp * Hp * sizeof(float), c->stream));
    CA(dalloc(&p.won, N)); CA(dalloc(&p.onsite, B));
    CA(hipMemsetAsync(p.won, 0, N * sizeof(float), c->stream));
    CA(hipMemsetAsync(p.onsite, 0, B * sizeof(float), c->stream));
    CA(dalloc(&p.woutp, Hp)); CA(dalloc(&p.bout, 1));
    if (c->split) CA(dalloc(&p.p16s, pack_split_dwords((int)NH)));
    CA(dalloc(&p.z1, B * Hp)); CA(dalloc(&p.logit, B)); CA(dalloc(&p.eloc, B));
    if (w == 0) {
      CA(dalloc(&p.z1_alt, B * Hp)); CA(dalloc(&p.logit_alt, B)); CA(dalloc(&p.onsite_alt, B));
      CA(hipMemsetAsync(p.onsite_alt, 0, B * sizeof(float), c->stream));
    }
  }
  CA(dalloc(&c->configs, B * N)); CA(dalloc(&c->configs_alt, B * N));
  CA(hipMemsetAsync(c->configs, 0, B * N * sizeof(float), c->stream));
  CA(hipMemsetAsync(c->configs_alt, 0, B * N * sizeof(float), c->stream));
  c->act.resize(L, nullptr);
  CA(dalloc(&c->act_all, L * B * Hp)); CA(dalloc(&c->act_alt, L * B * Hp));
  CA(hipMemsetAsync(c->act_all, 0, L * B * Hp * sizeof(float), c->stream));
  CA(hipMemsetAsync(c->act_alt, 0, L * B * Hp * sizeof(float), c->stream));
  {  // the sampler's stream outranks `stream`: where both have workgroups waiting for a CU, the sampler's go first
    int least = 0, greatest = 0;
    CA(hipDeviceGetStreamPriorityRange(&least, &greatest));
    CA(hipStreamCreateWithPriority(&c->sweep_stream, hipStreamNonBlocking, greatest));
  }
  CA(hipEventCreateWithFlags(&c->ev_mark, hipEventDisableTiming));
  CA(hipEventCreateWithFlags(&c->ev_now, hipEventDisableTiming));
  CA(hipEventCreateWithFlags(&c->ev_sweep_done, hipEventDisableTiming));
  for (int l = 0; l < L; ++l) c->act[l] = c->act_all + l * B * Hp;
  c->delta.resize(L, nullptr);
  CA(dalloc(&c->delta_all, L * B * Hp));
  CA(hipMemsetAsync(c->delta_all, 0, L * B * Hp * sizeof(float), c->stream));
  for (int l = 0; l < L; ++l) c->delta[l] = c->delta_all + l * B * Hp;
  for (int i = 0; i < 4; ++i) CA(hipMalloc(&c->d_batch[i / 2][i % 2], (size_t)(L + 1) * wgrad_problem_bytes()));
  {  // weight-gradient launch: the tiles of all layers
    // fully_connected on the kernels that run k_backprop16: the N = 1 output layer leaves the MFMA tile grid
    // (CGS_VMC_WGRAD_OUT_TILES=1 keeps it there: A/B measurements)
    const char* e_out = getenv("CGS_VMC_WGRAD_OUT_TILES");
    c->wg_out_partials = !rbm && !conv && !(wide && !c->wide_fast) && !(e_out && atoi(e_out) == 1);
    c->wg_tiles = plan_wgrad_total_tiles((int)N, c->H, (int)NH, rbm, !c->wg_out_partials);
    if (c->wg_out_partials) CA(dalloc(&c->wg_outpart, ((B + 15) / 16) * 2 * (Hp + 4)));
    CA(dalloc(&c->wg_tickets, c->wg_tiles > 0 ? c->wg_tiles : 1));
    CA(hipMemsetAsync(c->wg_tickets, 0, (size_t)(c->wg_tiles > 0 ? c->wg_tiles : 1) * sizeof(int), c->stream));
  }
  CA(dalloc(&c->ratio, B)); CA(dalloc(&c->ones, B)); CA(dalloc(&c->oscale, B));
  CA(launch_fill(c->stream, c->ones, 1.f, B));
  CA(launch_fill(c->stream, c->oscale, 1.f, B));
  if (c->hact == VMC_ACT_COS_) {
    CA(dalloc(&c->dact_all, L * B * Hp)); CA(dalloc(&c->dact_alt, L * B * Hp));
    CA(hipMemsetAsync(c->dact_all, 0, L * B * Hp * sizeof(float), c->stream));
    CA(hipMemsetAsync(c->dact_alt, 0, L * B * Hp * sizeof(float), c->stream));
  }
  CA(dalloc(&c->acc, 2 * P + 8)); CA(dalloc(&c->adam_m, P)); CA(dalloc(&c->adam_v, P));
  CA(dalloc(&c->grad_tmp, P));
  CA(hipMemsetAsync(c->acc, 0, (2 * P + 8) * sizeof(float), c->stream));
  CA(hipMemsetAsync(c->adam_m, 0, P * sizeof(float), c->stream));
  CA(hipMemsetAsync(c->adam_v, 0, P * sizeof(float), c->stream));
  CA(dalloc(&c->gemm_ws, plan_wgrad_ws_floats(c->wg_tiles, WG_MAX_SPLIT)));
  CA(dalloc(&c->d_accepted, 1)); CA(dalloc(&c->d_sum, 1)); CA(dalloc(&c->d_max, 1));
  CA(dalloc(&c->inj_up, B)); CA(dalloc(&c->inj_dn, B)); CA(dalloc(&c->inj_u, B));
  CA(dalloc(&c->acc_mask, B));
  CA(dalloc(&c->cnt, B)); CA(dalloc(&c->off, B + 1)); CA(dalloc(&c->diag, B));
  CA(dalloc(&c->cnt_alt, B)); CA(dalloc(&c->diag_alt, B));
  CA(dalloc(&c->rowinfo_id, B)); CA(launch_iota_rows(c->stream, c->rowinfo_id, (int)B));
  CA(dalloc(&c->bond_dummy, 1)); CA(hipMemsetAsync(c->bond_dummy, 0, sizeof(int2), c->stream));
  CA(dalloc(&c->offdiag, B));
  if (wide) {
    c->wrows = B > 131072 ? B : 131072;
    CA(dalloc(&c->wbuf[0], c->wrows * Hp)); CA(dalloc(&c->wbuf[1], c->wrows * Hp));
    CA(dalloc(&c->wide_u, B));
    CA(dalloc(&c->wide_dot, (long long)gemm_rowdot_tiles(c->H) * c->wrows));
    CA(dalloc(&c->wide_iup, B)); CA(dalloc(&c->wide_idn, B)); CA(dalloc(&c->wide_zero, Hp));
    CA(hipMemsetAsync(c->wide_zero, 0, Hp * sizeof(float), c->stream));
  }
  if (conv && c->conv_general) {
    // blocks of at most ~768 MB of im2col rows (one row configuration at least), at least B rows when that fits
    const long long per_row = (long long)cg.N * plan_cgen_lda(cg) * (long long)sizeof(float);
    long long block_mb = 768;
    if (const char* e = getenv("CGS_VMC_CONV_GENERAL_BLOCK_MB")) { block_mb = atoll(e); if (block_mb < 1) block_mb = 1; if (block_mb > 16384) block_mb = 16384; }
    long long rows = (block_mb << 20) / per_row;
    if (rows < 1) rows = 1;
    if (rows > (1LL << 30) / cg.N) rows = (1LL << 30) / cg.N;          // rows * N: the M of a GEMM (int)
    if (rows > (B > 65536 ? B : 65536)) rows = B > 65536 ? B : 65536;  // (small shapes: no point in hundreds of MB of maps)
    // whole rounds of the chip for the 128-row tiles of a block's products (the local energies run many full blocks:
    // 5.09 rounds of 256 CUs are paid as 6)
    if (rows * cg.N / 128 >= 2LL * c->num_cus) {
      const long long rounds = rows * cg.N / (128LL * c->num_cus);
      rows = rounds * 128LL * c->num_cus / cg.N;
    }
    if (const char* e = getenv("CGS_VMC_CONV_GENERAL_BLOCK_ROWS")) { const long long r = atoll(e); if (r >= 1 && r < rows) rows = r; }   // tests: several blocks at small shapes
    c->cg_rows = rows;
    // (cg_A, the im2col matrix: allocated by the first launch that writes one -- cgen_need_A, vmc_api_cgen.hip)
    // the band kernel (conv_band.hip) needs no im2col matrix: an untaped forward on it runs blocks sized by the two
    // maps alone, up to 2 GB of them (36 x 36 x 16 filters: 370 rows per im2col block against 12,900 -- the local
    // energies' 58 k rows were 110 blocks of partly filled launches)
    long long rows_fwd = rows;
    if (plan_cgen_band_ok(cg)) {
      const long long per_row_maps = 2LL * cg.N * cgen_fp(cg) * (long long)sizeof(float);
      rows_fwd = (2048LL << 20) / per_row_maps;
      if (rows_fwd > (1LL << 30) / cg.N) rows_fwd = (1LL << 30) / cg.N;
      if (rows_fwd > 131072) rows_fwd = 131072;
      if (rows_fwd < rows) rows_fwd = rows;
      if (getenv("CGS_VMC_CONV_GENERAL_BLOCK_ROWS")) rows_fwd = rows;      // tests: several blocks at small shapes
    }
    c->cg_rows_fwd = rows_fwd;
    for (int i = 0; i < 2; ++i) CA(dalloc(&c->cg_fm[i], rows_fwd * cg.N * cgen_fp(cg)));
    CA(dalloc(&c->cg_sum, rows_fwd)); CA(dalloc(&c->cg_zero, 1)); CA(dalloc(&c->cg_lnew, B));
    CA(hipMemsetAsync(c->cg_zero, 0, sizeof(float), c->stream));
    CA(dalloc(&c->wide_u, B)); CA(dalloc(&c->wide_iup, B)); CA(dalloc(&c->wide_idn, B));
  } else if (conv) {
    const long long nl = cg.n_conv > 1 ? cg.n_conv - 1 : 1;
    for (int w = 0; w < 2; ++w) {
      ParamSet& p = c->ps[w];
      CA(dalloc(&p.cw0, plan_conv_w0_floats(cg))); CA(dalloc(&p.cwf, plan_conv_wf_floats(cg)));
      CA(dalloc(&p.cwb, plan_conv_wf_floats(cg)));
      CA(dalloc(&p.cbias, plan_conv_bias_floats(cg)));
    }
    const int nw = conv_waves(cg);
    c->cG = conv_pick_group(cg, nw);
    if (c->cG > B) c->cG = (int)B;
    c->cGs = conv_pick_sweep_group(cg, B, c->num_cus, nw);     // chains per sampler workgroup (plan.hpp)
    if (const char* e = getenv("CGS_VMC_CONV_SWEEP_G")) {     // measurement knob: chains per sampler workgroup
      const int G = atoi(e);
      if (G >= 1 && G <= 64 && conv_rows_lds(cg, G) <= conv_lds_cap(cg)) c->cGs = G < B ? G : (int)B;
    }
    c->ctape_stride = B * cg.CS; c->cdelta_stride = B * cg.CS;
    CA(dalloc(&c->ctape, nl * c->ctape_stride)); CA(dalloc(&c->cdelta, (long long)cg.n_conv * c->cdelta_stride));
    c->c_slices = plan_conv_dw_slices(cg, B, c->num_cus);
    CA(dalloc(&c->cws, plan_conv_dw_ws_floats(cg, c->c_slices)));
  }
  CA(hipStreamSynchronize(c->stream));
#undef CA
  *out = c;
  return VMC_OK;
}

void vmc_destroy(vmc_ctx* c) {
  if (!c) return;
  DeviceGuard device_guard_(c->d.device);
  if (c->sweep_stream) hipStreamSynchronize(c->sweep_stream);
  hipStreamSynchronize(c->stream);
  drain_timings(c);
  if (c->sweep_stream) hipStreamDestroy(c->sweep_stream);
  for (hipStream_t q : c->cg_grp_stream) if (q) { hipStreamSynchronize(q); hipStreamDestroy(q); }
  for (hipEvent_t e : c->cg_grp_ev) if (e) hipEventDestroy(e);
  for (hipEvent_t e : {c->ev_mark, c->ev_now, c->ev_sweep_done}) if (e) hipEventDestroy(e);
  for (auto& e : c->event_pool) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
  for (int w = 0; w < 2; ++w) {
    ParamSet& p = c->ps[w];
    float* ptrs[] = {p.theta, p.w1p, p.b1p, p.bh, p.p16, p.p16t, p.woutp, p.bout, p.z1, p.logit, p.eloc, p.won, p.onsite,
                     p.z1_alt, p.logit_alt, p.onsite_alt, p.cw0, p.cwf, p.cwb, p.cbias};
    for (float* q : ptrs) if (q) hipFree(q);
    if (p.p16s) hipFree(p.p16s);
  }
  if (c->act_all) hipFree(c->act_all);
  if (c->act_alt) hipFree(c->act_alt);
  for (float* q : {c->oscale, c->dact_all, c->dact_alt, c->ctape, c->cdelta, c->cws, c->wbuf[0], c->wbuf[1],
                   c->wide_u, c->wide_zero}) if (q) hipFree(q);
  for (int* q : {c->wide_iup, c->wide_idn}) if (q) hipFree(q);
  if (c->wide_dot) hipFree(c->wide_dot);
  for (float* q : {c->cg_pmaps, c->cg_A, c->cg_fm[0], c->cg_fm[1], c->cg_zero, c->cg_lnew, c->cg_tape, c->cg_gl, c->cg_g[0], c->cg_g[1], c->cg_wpos,
                   c->cg_wt, c->cg_ws}) if (q) hipFree(q);
  if (c->cg_sum) hipFree(c->cg_sum);
  if (c->cg_td) hipFree(c->cg_td);
  if (c->cg_centre) hipFree(c->cg_centre);
  void* ptrs[] = {c->configs, c->configs_alt, c->bonds, c->half_jx, c->quarter_jz, c->cnt, c->off, c->diag, c->val,
                  c->offdiag, c->rowinfo, c->delta_all, c->d_batch[0][0], c->d_batch[0][1], c->d_batch[1][0], c->d_batch[1][1], c->ratio, c->ones, c->acc,
                  c->adam_m, c->adam_v, c->grad_tmp, c->gemm_ws, c->wg_tickets, c->d_accepted, c->d_sum,
                  c->d_max, c->tmp_cfg, c->tmp_z1, c->tmp_out, c->tmp_on, c->tmp_rowinfo, c->rowinfo_id, c->bond_dummy, c->inj_up, c->inj_dn, c->inj_u,
                  c->acc_mask, c->wg_outpart, c->cnt_alt, c->diag_alt};
  for (void* q : ptrs) if (q) hipFree(q);
  for (float* q : {c->sr_ctape, c->sr_cdelta, c->sr_cws, c->sr_cw0, c->sr_cwf, c->sr_cwb, c->sr_cbias}) if (q) hipFree(q);
  void* sr[] = {c->sr_cfg, c->sr_act, c->sr_delta, c->sr_ws, c->sr_t, c->sr_u, c->sr_x, c->sr_r,
                c->sr_p, c->sr_q, c->sr_partial, c->sr_sc, c->sr_ones, c->sr_tpart};
  for (void* q : sr) if (q) hipFree(q);
  if (c->h_stage) hipHostFree(c->h_stage);
  if (c->d_stage) hipFree(c->d_stage);
  if (c->d_eval) hipFree(c->d_eval);
  delete c;
}

int vmc_set_bonds(vmc_ctx* c, int32_t n_bonds, const int32_t* ij, const float* j_x, const float* j_z) {
  ENTER(c);
  if (n_bonds < 1 || !ij || !j_x || !j_z) return fail(c, VMC_ERR_INVALID, "bad bond arguments");
  // connected rows are indexed with 32-bit integers (at most one row per chain and bond)
  if ((long long)c->B * n_bonds > 0x7fffffffLL - c->B)
    return fail(c, VMC_ERR_UNSUPPORTED, "batch_size x n_bonds does not fit the 32-bit row index");
  std::vector<int2> b(n_bonds);
  std::vector<float> hx(n_bonds), qz(n_bonds);
  for (int k = 0; k < n_bonds; ++k) {
    const int i = ij[2 * k], j = ij[2 * k + 1];
    if (i < 0 || j < 0 || i >= c->N || j >= c->N || i == j)
      return fail(c, VMC_ERR_INVALID, "bond site index out of range (or i == j)");
    b[k] = make_int2(i, j);
    hx[k] = 0.5f * j_x[k];     // 0.25 * jx * 2   (operators.py:168-169)
    qz[k] = 0.25f * j_z[k];    // operators.py:169
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  void* old[] = {c->bonds, c->half_jx, c->quarter_jz, c->rowinfo, c->val};
  for (void* q : old) if (q) hipFree(q);
  c->bonds = nullptr; c->half_jx = c->quarter_jz = c->val = nullptr; c->rowinfo = nullptr;
  c->n_bonds = n_bonds;
  HIPCHK(c, dalloc(&c->bonds, n_bonds)); HIPCHK(c, dalloc(&c->half_jx, n_bonds));
  HIPCHK(c, dalloc(&c->quarter_jz, n_bonds));
  HIPCHK(c, dalloc(&c->rowinfo, (long long)c->B * n_bonds));
  HIPCHK(c, dalloc(&c->val, (long long)c->B * n_bonds));
  HIPCHK(c, hipMemcpy(c->bonds, b.data(), n_bonds * sizeof(int2), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->half_jx, hx.data(), n_bonds * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->quarter_jz, qz.data(), n_bonds * sizeof(float), hipMemcpyHostToDevice));
  c->list_valid = false;
  c->cnt_valid = false;
  return VMC_OK;
}

int vmc_set_params(vmc_ctx* c, int which, const float* theta) {
  ENTER(c);
  if ((which != 0 && which != 1) || !theta) return fail(c, VMC_ERR_INVALID, "bad arguments");
  HIPCHK(c, hipMemcpyAsync(c->ps[which].theta, theta, c->P * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->ps[which].has_params = true;
  c->ps[which].packed_valid = c->ps[which].cache_valid = false;
  if (which == VMC_PSI) c->acts_valid = false;
  return VMC_OK;
}

int vmc_get_params(vmc_ctx* c, int which, float* theta) {
  ENTER(c);
  if ((which != 0 && which != 1) || !theta) return fail(c, VMC_ERR_INVALID, "bad arguments");
  if (!c->ps[which].has_params) return fail(c, VMC_ERR_STATE, "parameters not set");
  HIPCHK(c, hipMemcpyAsync(theta, c->ps[which].theta, c->P * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return VMC_OK;
}

int vmc_transfer_params(vmc_ctx* c) {
  ENTER(c);
  if (!c->ps[0].has_params) return fail(c, VMC_ERR_STATE, "parameters not set");
  HIPCHK(c, hipMemcpyAsync(c->ps[1].theta, c->ps[0].theta, c->P * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  c->ps[1].has_params = true;
  c->ps[1].packed_valid = c->ps[1].cache_valid = false;
  return VMC_OK;
}

int vmc_set_configs(vmc_ctx* c, const float* configs) {
  ENTER(c);
  if (!configs) return fail(c, VMC_ERR_INVALID, "null configs");
  const long long n = (long long)c->B * c->N;
  // staged in the alternate chain buffer (free between sampler launches), validated on the
  // device, and only then made current: a rejected batch leaves the chains untouched
  HIPCHK(c, hipMemcpyAsync(c->configs_alt, configs, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, launch_check_pm1(c->stream, c->configs_alt, n, c->cnt));
  int bad = 0;
  HIPCHK(c, hipMemcpyAsync(&bad, c->cnt, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (bad) return fail(c, VMC_ERR_INVALID, "configs must be +-1");
  HIPCHK(c, hipMemcpyAsync(c->configs, c->configs_alt, n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  invalidate_configs(c);
  return VMC_OK;
}

int vmc_get_configs(vmc_ctx* c, float* configs) {
  ENTER(c);
  if (!configs) return fail(c, VMC_ERR_INVALID, "null configs");
  HIPCHK(c, hipMemcpyAsync(configs, c->configs, (long long)c->B * c->N * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return VMC_OK;
}

int vmc_set_shift(vmc_ctx* c, int which, float shift) {
  CHECK_CTX(c);
  if (which != 0 && which != 1) return fail(c, VMC_ERR_INVALID, "bad which");
  c->ps[which].shift = shift;
  return VMC_OK;
}

int vmc_get_shift(vmc_ctx* c, int which, float* shift) {
  CHECK_CTX(c);
  if ((which != 0 && which != 1) || !shift) return fail(c, VMC_ERR_INVALID, "bad arguments");
  *shift = c->ps[which].shift;
  return VMC_OK;
}

int vmc_amplitude(vmc_ctx* c, int which, const float* configs, int64_t n_rows, float* logit, float* psi) {
  ENTER(c);
  if (which != 0 && which != 1) return fail(c, VMC_ERR_INVALID, "bad which");
  if (n_rows < 0) return fail(c, VMC_ERR_INVALID, "n_rows < 0");
  std::vector<float> host((size_t)n_rows);
  if (!configs) {
    if (n_rows != c->B) return fail(c, VMC_ERR_INVALID, "n_rows must equal batch_size when configs == NULL");
    PROPAGATE(ensure_cache(c, which));
    HIPCHK(c, hipMemcpyAsync(host.data(), c->ps[which].logit, n_rows * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  } else if (n_rows > 0) {
    PROPAGATE(ensure_packed(c, which));
    PROPAGATE(grow_tmp(c, n_rows));
    HIPCHK(c, hipMemcpyAsync(c->tmp_cfg, configs, n_rows * c->N * sizeof(float), hipMemcpyHostToDevice, c->stream));
    ParamSet& p = c->ps[which];
    if (c->conv) {
      PROPAGATE(conv_rows(c, which, c->tmp_cfg, c->tmp_rowinfo, (int)n_rows, nullptr, false, c->tmp_out, false));
    } else if (c->wide) {
      PROPAGATE(first_layer(c, p, c->tmp_cfg, c->tmp_z1, (int)n_rows));
      if (c->rbm) HIPCHK(c, launch_onsite(c->stream, c->tmp_cfg, p.won, (int)n_rows, c->N, c->tmp_on));
      PROPAGATE(wide_forward(c, which, c->tmp_z1, c->tmp_rowinfo, n_rows, false, c->tmp_out, c->tmp_on));
    } else {
      PROPAGATE(first_layer(c, p, c->tmp_cfg, c->tmp_z1, (int)n_rows));
      if (c->rbm) HIPCHK(c, launch_onsite(c->stream, c->tmp_cfg, p.won, (int)n_rows, c->N, c->tmp_on));
      TailArgs a = tail_args(c, which);
      a.z1 = c->tmp_z1; a.on_base = c->tmp_on; a.n_rows = (int)n_rows; a.out = c->tmp_out; a.rowinfo = c->tmp_rowinfo;
      HIPCHK(c, launch_rows(c, which, a, false));
    }
    HIPCHK(c, hipMemcpyAsync(host.data(), c->tmp_out, n_rows * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const float shift = c->ps[which].shift;
  for (int64_t i = 0; i < n_rows; ++i) {
    if (logit) logit[i] = host[i];
    // wavefunctions.py:350-353: exp(x - shift) (232), or the output activation itself, no shift
    if (psi) psi[i] = c->oact == VMC_ACT_EXP_ ? expf(host[i] - shift) : host_activation(c->oact, host[i]);
  }
  return VMC_OK;
}

int vmc_local_energy(vmc_ctx* c, int which, float* eloc, double* mean) {
  ENTER(c);
  if (which != 0 && which != 1) return fail(c, VMC_ERR_INVALID, "bad which");
  PROPAGATE(local_energy_device(c, which));
  if (mean) HIPCHK(c, launch_sum(c->stream, c->ps[which].eloc, c->B, c->d_sum));
  if (eloc) HIPCHK(c, hipMemcpyAsync(eloc, c->ps[which].eloc, c->B * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  int cnt_total = 0;
  HIPCHK(c, hipMemcpyAsync(&cnt_total, c->off + c->B, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  double s = 0.0;
  if (mean) HIPCHK(c, hipMemcpyAsync(&s, c->d_sum, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->last_rows = cnt_total;
  if (mean) *mean = s / (double)c->B;
  return VMC_OK;
}

int vmc_local_energy_terms(vmc_ctx* c, int which, float* diag, float* offdiag_over_psi) {
  ENTER(c);
  if (which != 0 && which != 1) return fail(c, VMC_ERR_INVALID, "bad which");
  PROPAGATE(local_energy_device(c, which));
  if (diag) HIPCHK(c, hipMemcpyAsync(diag, c->diag, c->B * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (offdiag_over_psi) HIPCHK(c, hipMemcpyAsync(offdiag_over_psi, c->offdiag, c->B * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return VMC_OK;
}

int vmc_debug_kernel_path(vmc_ctx* c, int32_t* path) {
  CHECK_CTX(c);
  if (!path) return fail(c, VMC_ERR_INVALID, "null");
  *path = c->conv ? (c->conv_general ? 6 : 3) : (c->wide ? (c->wide_fast ? 1 : 2) : (c->split ? (c->split_sweep ? 5 : 4) : 0));
  return VMC_OK;
}

int vmc_debug_sweep_tile(vmc_ctx* c, int32_t set, int32_t* chains) {
  ENTER(c);
  if (set != 0 && set != 8 && set != 16) return fail(c, VMC_ERR_INVALID, "sweep tile: 0 (query), 8 or 16");
  if (set == 8 && !c->sweep8_ok) return fail(c, VMC_ERR_UNSUPPORTED, "no eight-chain sampler for this shape (fully_connected + relu, 128 or 256 padded units, n_sites <= units)");
  if (set) { PROPAGATE(join_sweep(c)); c->sweep_tile = set; }
  if (chains) *chains = c->sweep_tile;
  return VMC_OK;
}

int vmc_last_connected_rows(vmc_ctx* c, int64_t* rows) { CHECK_CTX(c); if (!rows) return fail(c, VMC_ERR_INVALID, "null"); *rows = c->last_rows; return VMC_OK; }

int vmc_timing_enable(vmc_ctx* c, int on) {
  CHECK_CTX(c);
  c->timing = on < 0 || on > 2 ? 1 : on;
  while (c->timing && c->event_pool.size() < 64) {   // enough for a dozen steps in flight
    hipEvent_t a, b;
    HIPCHK(c, hipEventCreate(&a)); HIPCHK(c, hipEventCreate(&b));
    c->event_pool.emplace_back(a, b);
  }
  return VMC_OK;
}

int vmc_timing_reset(vmc_ctx* c) {
  CHECK_CTX(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  drain_timings(c);
  c->timings.clear();
  return VMC_OK;
}

int vmc_timing_get(vmc_ctx* c, const char* name, double* ms, int64_t* launches) {
  CHECK_CTX(c);
  if (!name) return fail(c, VMC_ERR_INVALID, "null name");
  drain_timings(c);
  auto it = c->timings.find(name);
  if (ms) *ms = it == c->timings.end() ? 0.0 : it->second.first;
  if (launches) *launches = it == c->timings.end() ? 0 : it->second.second;
  return VMC_OK;
}

int vmc_synchronize(vmc_ctx* c) {
  ENTER(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipStreamSynchronize(c->sweep_stream));
  return VMC_OK;
}

int vmc_debug_gemm(vmc_ctx* c, int32_t M, int32_t N, int32_t K, const float* A, int64_t sam, int64_t sak,
                   int64_t a_len, const float* B, int64_t sbk, int64_t sbn, int64_t b_len, float* C) {
  ENTER(c);
  float *dA = nullptr, *dB = nullptr, *dC = nullptr, *ws = nullptr;
  HIPCHK(c, dalloc(&dA, a_len)); HIPCHK(c, dalloc(&dB, b_len)); HIPCHK(c, dalloc(&dC, (long long)M * N));
  HIPCHK(c, dalloc(&ws, 4LL * M * N));
  HIPCHK(c, hipMemcpy(dA, A, a_len * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(dB, B, b_len * sizeof(float), hipMemcpyHostToDevice));
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.A = dA; g.sam = sam; g.sak = sak; g.B = dB; g.sbk = sbk; g.sbn = sbn;
  g.M = M; g.N = N; g.K = K; g.C = dC; g.ldc = N; g.epilogue = 0;
  g.splitk = K >= 256 ? 4 : 1; g.workspace = ws;
  HIPCHK(c, launch_gemm(c->stream, g));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(C, dC, (long long)M * N * sizeof(float), hipMemcpyDeviceToHost));
  hipFree(dA); hipFree(dB); hipFree(dC); hipFree(ws);
  return VMC_OK;
}


}  // extern "C"
