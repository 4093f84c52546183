// Host side of libcgsvmc_hip.so: the C ABI of include/cgsvmc.h on top of the gfx950
// kernels in mlp.hip / eloc.hip / grad.hip.  One vmc_ctx per GPU, all work on ctx->stream.
#include "../../include/cgsvmc.h"
#include "common.hpp"
#include "conv.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace {

std::string g_create_error;

struct ParamSet {
  float* theta = nullptr;
  float *w1p = nullptr, *b1p = nullptr, *bh = nullptr, *p16 = nullptr, *p16t = nullptr,
        *woutp = nullptr, *bout = nullptr, *won = nullptr;
  float* z1 = nullptr;     // [B][Hp] cache for the ctx's chains
  float* onsite = nullptr; // [B] cached x . w_on (RBM)
  float* logit = nullptr;  // [B]
  // psi only: the buffers the NEXT sampler launch writes (see vmc_ctx::configs_alt)
  float *z1_alt = nullptr, *onsite_alt = nullptr, *logit_alt = nullptr;
  float* eloc = nullptr;   // [B]
  // convolutional ansatz types: fragment images of conv.hpp ConvParams
  float *cw0 = nullptr, *cwf = nullptr, *cwb = nullptr, *cbias = nullptr;
  unsigned* p16s = nullptr;   // CGS_VMC_SPLIT_BF16=1: the H x H layers as three bf16 terms (tail_split.hip)
  bool packed_valid = false, cache_valid = false, has_params = false;
  float shift = -10.f;     // wavefunctions.py:209
  PackedParams packed() const { return PackedParams{w1p, b1p, bh, p16, woutp, bout, won}; }
};

struct TimedRegion {
  std::string name;
  hipEvent_t start, stop;
};

}  // namespace

struct vmc_ctx {
  vmc_desc d;
  int N = 0, B = 0, L = 0, H = 0, Hp = 0;
  bool rbm = false;        // RestrictedBoltzmannNetwork instead of FullyConnectedNetwork
  // Conv2DNetwork / ResNet2D (conv.hip).  The dense-ansatz members below keep harmless minimal
  // shapes (H = filters, Hp = 64, no H x H layer); acts_valid tells whether the forward tapes
  // hold the inputs of every convolution for psi on the current chains.
  bool conv = false;
  ConvGeom cg;
  int cG = 1, cGs = 1;     // samples per workgroup pass of the row / backward kernels, of the sampler
  float *ctape = nullptr, *cdelta = nullptr, *cws = nullptr;
  long long ctape_stride = 0, cdelta_stride = 0;
  int c_slices = 64;       // sample slices of the weight-gradient kernel
  // fully_connected with more than 256 hidden units: general path (wide.hip)
  bool wide = false;
  // ... except relu networks of at most 512 units with an H x H layer: their sampler and row kernel
  // are instantiations of the fused kernels (k_sweep16<24|32>, k_tail_lds); only the gradient path
  // stays on the general GEMMs.  CGS_VMC_WIDE_FAST=0 forces the general path.
  bool wide_fast = false;
  // EXPERIMENT (CGS_VMC_SPLIT_BF16=1; fully_connected, relu, 193 .. 256 units, >= 1 H x H layer): the row
  // kernel computes its fp32 results on the bf16 matrix cores from three-term splits (tail_split.hip)
  bool split = false;             // CGS_VMC_SPLIT_BF16 >= 1: the row kernel on the BF16 matrix cores (3 x bf16 split, EXPERIMENT)
  bool split_sweep = false;       // CGS_VMC_SPLIT_BF16 == 2: the sampler's H x H layers too (k_sweep16s)
  long long wrows = 0;     // rows of the two activation row buffers
  float *wbuf[2] = {nullptr, nullptr}, *wide_u = nullptr, *wide_zero = nullptr;
  double* wide_dot = nullptr;          // [ceil(H / 128)][wrows] row-dot partials of the last H x H layer (GemmArgs epilogue 10)
  // general convolution path (conv_general.hip; plan.hpp: conv beyond the fused kernels' limits): block buffers
  bool conv_general = false;
  long long cg_rows = 0;                   // row configurations per block
  float* cg_A = nullptr;                   // im2col rows [cg_rows * N][plan_cgen_lda]
  float* cg_fm[2] = {nullptr, nullptr};    // feature maps [cg_rows][N][Fp] (cgen_post: activations; the cosine: pre-activations)
  double* cg_sum = nullptr;                // [cg_rows] sums of the last map
  float* cg_zero = nullptr;                // one 0.f (the "b_out" of wide_out_finish)
  float* cg_lnew = nullptr;                // [B] candidate logits of the sampler
  // ... its gradient path (allocated by the first gradient call): the map of every convolution (the tape), two
  // d logit / d map buffers, per-position weights, the transposed weight images, the split-K workspace
  float* cg_tape = nullptr; float* cg_gl = nullptr; float* cg_g[2] = {nullptr, nullptr}; float* cg_wpos = nullptr; float* cg_wt = nullptr;
  float* cg_ws = nullptr; long long cg_ws_floats = 0;
  double* cg_td = nullptr;                 // [cg_rows] O_b . v of a block (SR)
  float* cg_centre = nullptr;              // [1] mean of O_b . v over the stored samples (SR)
  bool sr_centre = false;                  // the SR matvec may centre its weights: a single-rank solve is running
  bool sr_phase1_done = false;             // vmc_sr_matvec_phase1 has run for the current CG direction (general convolution path)
  int *wide_iup = nullptr, *wide_idn = nullptr;
  int hact = VMC_ACT_RELU_;  // hidden activation (layers.NONLINEARITIES id)
  int oact = VMC_ACT_EXP_;   // output activation; exp: psi = exp(x - shift), else psi = g(x), no shift
  float* oscale = nullptr;   // [B] (1/psi) d psi / d x of a non-exp output activation
  float *dact_all = nullptr, *dact_alt = nullptr;   // [L][B][Hp] f'(z) next to act_all (cosine only)
  int n_hh = 0;            // H x H layers: L - 1 (FC) or L (RBM)
  int A = 0;               // activation buffers = n_hh + 1
  ParamLayout lay;
  long long P = 0;
  hipStream_t stream = nullptr;
  ParamSet ps[2];
  float* configs = nullptr;
  // Double-buffered chain state.  A sampler launch reads {configs, z1, logit} and writes
  // {configs_alt, z1_alt, logit_alt, onsite_alt, act_alt}; the two sets are swapped on the host
  // right after the launch.  accumulate(R_t) on `stream` and sweep(R_t -> R_t+1) on
  // `sweep_stream` therefore touch disjoint buffers and run concurrently (training.py:614-617:
  // the two ops of a batch iteration are independent given the chains R_t).
  float* configs_alt = nullptr;
  float* act_alt = nullptr;
  int parity = 0;                 // which physical buffer set is current (GEMM tables are per set)
  hipStream_t sweep_stream = nullptr;   // private non-blocking stream of the sampler
  bool overlap = true;            // CGS_VMC_OVERLAP=0: everything on `stream`
  bool overlap_full = false;      // CGS_VMC_OVERLAP=2: overtake even when the sampler fills every CU
  bool side_sweep_once = false;   // the next vmc_mc_steps goes to sweep_stream BEHIND everything enqueued so far, so that
                                  // what follows on `stream` (the accumulator all-reduce of a sharded epoch) runs beside it
  hipEvent_t ev_mark = nullptr;   // recorded on `stream` at the start of the latest accumulate
  hipEvent_t ev_now = nullptr;    // scratch: "everything enqueued on `stream` so far"
  hipEvent_t ev_sweep_done = nullptr;
  bool sweep_pending = false;     // a sampler launch on sweep_stream that `stream` has not waited for
  bool token = false;             // the latest entry point was an accumulate the next sweep may overtake
  bool expect_sweep = false;      // the previous accumulate was overtaken by a sweep: leave it CUs
  bool acc_since_sweep = false;   // a gradient accumulate may follow: the sampler hands over activations
  // Hamiltonian
  int n_bonds = 0;
  int2* bonds = nullptr;
  float *half_jx = nullptr, *quarter_jz = nullptr;
  int *cnt = nullptr, *off = nullptr;
  float *diag = nullptr, *val = nullptr, *offdiag = nullptr;
  int2* rowinfo = nullptr;
  int2* bond_dummy = nullptr;   // {0,0}: stands in for the bond table before vmc_set_bonds
  int2* rowinfo_id = nullptr;   // identity list {r, 0} for plain rows (cache refresh)
  int2* tmp_rowinfo = nullptr;
  bool list_valid = false;
  bool cnt_valid = false;          // cnt / diag hold the census of `configs` (left by the sampler's last launch)
  int* cnt_alt = nullptr; float* diag_alt = nullptr;   // the census the NEXT sampler launch writes (swapped with the chains)
  long long last_rows = 0;
  // gradient path
  std::vector<float*> act;   // L views [B][Hp] into act_all
  float* act_all = nullptr;  // [L][B][Hp]
  bool acts_valid = false;   // act[] hold the activations of psi on the current chains
  std::vector<float*> delta;   // L views [B][Hp] into delta_all: d logit / d z_l
  float* delta_all = nullptr;
  void* d_batch[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // weight-gradient problem tables [w = eloc / ratio][parity]
  bool batch_ready[2][2] = {{false, false}, {false, false}};
  int wg_tiles = 0;                // MFMA tiles of the weight-gradient launch (plan.hpp)
  bool wg_out_partials = false;    // the output layer's sums come from k_backprop16's partials (OutLayerSums)
  float* wg_outpart = nullptr;     // [ceil(B / 16)][2][Hp + 4]
  int* wg_tickets = nullptr;       // [wg_tiles] arrival tickets of the split-K fold, zero between launches
  float *ratio = nullptr, *ones = nullptr;
  float *acc = nullptr, *adam_m = nullptr, *adam_v = nullptr, *grad_tmp = nullptr;
  // reset_gradients does not zero `acc` at once: the first dense accumulate after it WRITES its sums
  // (no 1.3 MB memset + read-modify-write per optimizer step); everything else that touches `acc`
  // materialises the zeros first (acc_zeros)
  bool acc_fresh = false;
  long long adam_t = 0;
  float* gemm_ws = nullptr;  // partial tiles of the weight-gradient launch: plan_wgrad_ws_floats(wg_tiles, WG_MAX_SPLIT)
  int num_cus = 256;
  int sweep_waves = 8;       // waves per sweep workgroup at Hp = 256 (CGS_VMC_SWEEP_WAVES=4|8)
  int sweep_no_w1l = 0;      // CGS_VMC_SWEEP_W1L=0: W1 stays in L2 (smaller LDS footprint)
  int sweep_tile = 16;       // chains per sampler workgroup: 16 (k_sweep16) or 8 (k_sweep8; plan_sweep_tile)
  bool sweep8_ok = false;    // the shape has a k_sweep8
  // stochastic reconfiguration (extension, sr.hip): sample store + CG vectors
  int sr_cap = 0, sr_n = 0, sr_iter = 0;
  float *sr_cfg = nullptr, *sr_act = nullptr, *sr_delta = nullptr;   // [cap B][N], [L][cap B][Hp] x2
  // convolutional ansatz types: stored tapes / deltas [n_conv-1 | n_conv][cap B][CS], the CG direction
  // packed like a parameter set, and the slices of the weight-gradient kernel over the stored samples
  float *sr_ctape = nullptr, *sr_cdelta = nullptr, *sr_cws = nullptr;
  float *sr_cw0 = nullptr, *sr_cwf = nullptr, *sr_cwb = nullptr, *sr_cbias = nullptr;
  int sr_cslices = 0;
  float *sr_ws = nullptr, *sr_t = nullptr, *sr_ones = nullptr;        // [slices][(max(N,H)+1) H], [cap B] x2
  float* sr_tpart = nullptr;                                          // [layers x column blocks][cap B] partial t
  float *sr_u = nullptr, *sr_x = nullptr, *sr_r = nullptr, *sr_p = nullptr, *sr_q = nullptr;
  double *sr_partial = nullptr, *sr_sc = nullptr;
  bool sr_begun = false;
  // collectives over sharded chains: host hook for non-RCCL transports + its staging
  vmc_host_allreduce_fn host_reduce = nullptr;
  void* host_reduce_user = nullptr;
  int host_reduce_caps = 0;                       // VMC_HOST_REDUCE_CAP_*: what the registered host hook has declared
  vmc_device_allreduce_fn dev_reduce = nullptr;   // in-stream transport of the host's own collective library
  void* dev_reduce_user = nullptr;
  double* d_eval = nullptr;      // vmc_evaluate: batch sums / means of the samples
  int d_eval_n = 0;
  float* h_stage = nullptr;      // pinned
  float* d_stage = nullptr;      // vmc_debug_allreduce only
  long long h_stage_n = 0, d_stage_n = 0;
  // scratch
  unsigned long long* d_accepted = nullptr;
  double* d_sum = nullptr;
  float* d_max = nullptr;
  float *tmp_cfg = nullptr, *tmp_z1 = nullptr, *tmp_out = nullptr, *tmp_on = nullptr;
  long long tmp_rows = 0;
  int *inj_up = nullptr, *inj_dn = nullptr;
  float* inj_u = nullptr;
  unsigned char* acc_mask = nullptr;
  unsigned long long step = 0;
  // timing
  int timing = 0;            // 0 off, 1 every region, 2 the two roofline kernels only
  std::vector<TimedRegion> pending;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> event_pool;
  std::map<std::string, std::pair<double, long long>> timings;
  std::string err;
};

namespace {

int fail(vmc_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg; else g_create_error = msg;
  return code;
}

#define HIPCHK(c, expr)                                                                  \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess)                                                                \
      return fail((c), VMC_ERR_HIP,                                                      \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                    \
  } while (0)

// Every entry point runs on the ctx's device whatever the calling thread's current device is
// (HIP's current device is per thread), and restores the caller's device on return.
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != dev && hipSetDevice(dev) == hipSuccess) prev = cur;
  }
  ~DeviceGuard() { if (prev >= 0) hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// entry points that only touch the accumulators / scalars / host state
#define CHECK_CTX(c)                                                      \
  if (!(c)) return fail(nullptr, VMC_ERR_INVALID, "null ctx");            \
  DeviceGuard device_guard_((c)->d.device)

// every other entry point: the work it enqueues on `stream` may depend on the chains, so
// `stream` first waits for a sampler launch still in flight on sweep_stream
#define ENTER(c)                                                          \
  CHECK_CTX(c);                                                           \
  (c)->token = false;                                                     \
  do { int rc_join_ = join_sweep(c); if (rc_join_ != VMC_OK) return rc_join_; } while (0)

#define PROPAGATE(expr) \
  do { int rc_ = (expr); if (rc_ != VMC_OK) return rc_; } while (0)

// CUs a sampler launch occupies (8 waves at 255 registers, or LDS, fill a CU per workgroup)
int sweep_cus(const vmc_ctx* c) { return c->sweep_tile == 8 ? (c->B + 7) / 8 : (c->B + 15) / 16; }

// The sampler may overtake the accumulate enqueued just before it when it leaves the local-energy
// kernel at least a quarter of the CUs; with one 16-chain tile per CU (config 3) there is nothing
// to share and the launch stays on `stream`.
bool can_overlap(const vmc_ctx* c) {
  return c->overlap && (c->overlap_full || sweep_cus(c) <= (3 * c->num_cus) / 4);
}

// `acc` is about to be read or partially written: turn a pending reset into real zeros
int acc_zeros(vmc_ctx* c) {
  if (c->acc_fresh) {
    hipError_t e = hipMemsetAsync(c->acc, 0, (2 * c->P + 8) * sizeof(float), c->stream);
    if (e != hipSuccess) return fail(c, VMC_ERR_HIP, std::string("hipMemsetAsync: ") + hipGetErrorString(e));
    c->acc_fresh = false;
  }
  return VMC_OK;
}

int join_sweep(vmc_ctx* c) {
  if (c->sweep_pending) {
    hipError_t e = hipStreamWaitEvent(c->stream, c->ev_sweep_done, 0);
    if (e != hipSuccess) return fail(c, VMC_ERR_HIP, std::string("hipStreamWaitEvent: ") + hipGetErrorString(e));
    c->sweep_pending = false;
  }
  return VMC_OK;
}

void swap_chain_buffers(vmc_ctx* c) {
  ParamSet& p = c->ps[0];
  std::swap(c->configs, c->configs_alt);
  std::swap(p.z1, p.z1_alt); std::swap(p.logit, p.logit_alt); std::swap(p.onsite, p.onsite_alt);
  std::swap(c->act_all, c->act_alt);
  std::swap(c->dact_all, c->dact_alt);
  std::swap(c->cnt, c->cnt_alt); std::swap(c->diag, c->diag_alt);
  for (size_t l = 0; l < c->act.size(); ++l) c->act[l] = c->act_all + (long long)l * c->B * c->Hp;
  c->parity ^= 1;
}

template <typename T>
hipError_t dalloc(T** p, long long n) {
  return hipMalloc((void**)p, (size_t)(n > 0 ? n : 1) * sizeof(T));
}

// Per-kernel timing: event pairs come from a pool (creating two events per region costs more
// than recording them); regions whose stop event has completed are folded into the totals and
// their events recycled without blocking.
static void account(vmc_ctx* c, const TimedRegion& r) {
  float ms = 0.f;
  hipEventElapsedTime(&ms, r.start, r.stop);
  auto& t = c->timings[r.name];
  t.first += ms; t.second += 1;
  c->event_pool.emplace_back(r.start, r.stop);
}

static void harvest_finished(vmc_ctx* c) {
  size_t done = 0;
  while (done < c->pending.size() && hipEventQuery(c->pending[done].stop) == hipSuccess) {
    account(c, c->pending[done]);
    ++done;
  }
  if (done) c->pending.erase(c->pending.begin(), c->pending.begin() + done);
}

struct Timer {
  vmc_ctx* c; hipStream_t st; bool on; TimedRegion r;
  Timer(vmc_ctx* ctx, const char* name, hipStream_t stream = nullptr, bool own_stream = false)
      : c(ctx), st(own_stream ? stream : ctx->stream), on(ctx->timing == 1 || (ctx->timing == 2 && (!strcmp(name, "sweep") || !strcmp(name, "tail_eloc")))) {
    if (on) {
      r.name = name;
      if (c->event_pool.empty()) harvest_finished(c);
      if (c->event_pool.empty()) {
        hipEventCreate(&r.start); hipEventCreate(&r.stop);
      } else {
        r.start = c->event_pool.back().first; r.stop = c->event_pool.back().second;
        c->event_pool.pop_back();
      }
      hipEventRecord(r.start, st);
    }
  }
  ~Timer() {
    if (on) { hipEventRecord(r.stop, st); c->pending.push_back(r); }
  }
};

void drain_timings(vmc_ctx* c) {
  for (auto& r : c->pending) {
    hipEventSynchronize(r.stop);
    account(c, r);
  }
  c->pending.clear();
}

long long off_w(const vmc_ctx* c, int l) { return plan_off_w(c->lay, c->H, l); }   // weight matrix of layer l (0 = first)
long long off_b(const vmc_ctx* c, int l) { return plan_off_b(c->lay, c->H, l); }   // biases sit right behind their weights
long long off_wout(const vmc_ctx* c) { return c->lay.off_wout; }
long long off_bout(const vmc_ctx* c) { return c->lay.off_bout; }

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
int ensure_cache(vmc_ctx* c, int which);
#define CGEN_SPLITK 32     // K slices of the weight-gradient products of the general convolution path

// One convolution of the general path over `rows` row configurations: im2col gather of its input into cg_A, then the
// product with the parameter slice; residual: dst += (ResBlock2d's `v + h`).
// What a stored map of the general path holds: the ACTIVATION of a convolution's output (conv_plain: f(z_l) behind every
// convolution but the last; residual blocks: selu(u) behind a block's first convolution, the linear h elsewhere) -- so
// that the next convolution can gather it as it stands -- unless the hidden activation is the cosine, whose derivative
// needs the pre-activation: then the map holds z_l and f is applied on the gather (as in the first form of this path).
static bool cgen_post(const vmc_ctx* c) { return c->cg.resnet || c->cg.hact != VMC_ACT_COS_; }
static int cgen_in_pre(const vmc_ctx* c, int l) {       // activation applied while convolution l's input is gathered
  if (l == 0 || c->cg.resnet || cgen_post(c)) return -1;
  return c->cg.hact;
}
static bool cgen_implicit_on() {
  static const bool on = !(getenv("CGS_VMC_CONV_GENERAL_IMPLICIT") && atoi(getenv("CGS_VMC_CONV_GENERAL_IMPLICIT")) == 0);
  return on;
}

// CGS_VMC_CONV_BAND=0: the im2col + GEMM form for every filter count (read per call: A/B tests in one process)
static bool cgen_band_on() { const char* e = getenv("CGS_VMC_CONV_BAND"); return !(e && atoi(e) == 0); }

static int cgen_conv(vmc_ctx* c, const ParamSet& p, const float* configs, const int2* rowinfo, const int* iup,
                     const int* idn, int l, int rows, const float* in, float* dst, long long row0) {
  const ConvGeom& g = c->cg;
  const int Fp = cgen_fp(g), lda = plan_cgen_lda(g);
  GemmArgs m; memset(&m, 0, sizeof(m));
  m.B = p.theta + cgen_off_w(g, l); m.sbk = g.F; m.sbn = 1;
  m.M = rows * g.N; m.N = g.F; m.K = cgen_kdim(g, l); m.C = dst; m.ldc = Fp;
  m.bias = p.theta + cgen_off_b(g, l); m.splitk = 1;
  if (!g.resnet) { m.epilogue = (l + 1 < g.n_conv && cgen_post(c)) ? 1 : 4; m.act = g.hact; }
  else m.epilogue = l == 0 ? 4 : ((l & 1) ? 11 : 8);       // initial convolution; selu(first_conv(h)); h + second_conv(.)
  const int pre = cgen_in_pre(c, l);
  // up to 16 filters: the band kernel (conv_band.hip) -- no im2col matrix, no 64-column tile for 16 columns
  const bool first_direct = l == 0 && cgen_band_on() && cgen_first_direct_ok(g, m.epilogue);
  if (cgen_band_on() && (cgen_band_ok(g) || first_direct)) {
    CgenBandArgs b; memset(&b, 0, sizeof(b));
    b.g = g; b.layer = l; b.Fp = Fp; b.w = m.B; b.bias = m.bias; b.in = in; b.out = dst; b.rows = rows;
    b.pre_act = pre; b.epilogue = m.epilogue; b.act = m.act;
    if (l == 0) {
      b.configs = configs; b.rowinfo = rowinfo; b.row0 = row0; b.bonds = c->bonds ? c->bonds : c->bond_dummy;
      b.iup = iup; b.idn = idn;
    }
    if (cgen_band_ok(g)) HIPCHK(c, launch_cgen_band(c->stream, b, c->num_cus));
    else HIPCHK(c, launch_cgen_first_direct(c->stream, b, c->num_cus));     // more than 16 filters: the first convolution only
    return VMC_OK;
  }
  // the gather inside the product's A operand (k_gemm_ring<., true>): no im2col matrix for this convolution
  if (l > 0 && pre < 0 && cgen_implicit_on()) {
    m.A = in; m.conv_a = 1; m.ca_N = g.N; m.ca_D1 = g.D1; m.ca_D2 = g.D2; m.ca_KW = g.KW; m.ca_lo = g.lo; m.ca_lo2 = g.lo2;
    m.ca_F = g.F; m.ca_Fp = Fp;
    if (gemm_conv_a_ok(m)) { HIPCHK(c, launch_gemm(c->stream, m)); return VMC_OK; }
    m.conv_a = 0;
  }
  CgenIm2colArgs a;
  memset(&a, 0, sizeof(a));
  a.g = g; a.layer = l; a.Fp = Fp; a.pre_act = pre; a.rows = rows; a.lda = lda; a.A = c->cg_A;
  if (l == 0) {
    a.src = configs; a.rowinfo = rowinfo; a.row0 = row0; a.bonds = c->bonds ? c->bonds : c->bond_dummy;
    a.iup = iup; a.idn = idn;
  } else {
    a.src = in;
  }
  HIPCHK(c, launch_cgen_im2col(c->stream, a));
  m.A = c->cg_A; m.sam = lda; m.sak = 1;
  HIPCHK(c, launch_gemm(c->stream, m));
  return VMC_OK;
}

// tape != nullptr (gradient path, n_rows <= cg_rows): the map of convolution l is kept at tape + l * tape_stride
// (cgen_post says what it holds; for the second convolution of a residual block the block's output h + v)
static int cgen_forward(vmc_ctx* c, int which, const float* configs, const int2* rowinfo, long long n_rows,
                        const int* iup, const int* idn, bool ratio, float* out, float* tape = nullptr,
                        long long tape_stride = 0, long long first_row = 0) {
  const ConvGeom& g = c->cg;
  const ParamSet& p = c->ps[which];
  const int Fp = cgen_fp(g);
  auto conv = [&](int l, int rows, const float* in, float* dst, long long row0) -> int {
    return cgen_conv(c, p, configs, rowinfo, iup, idn, l, rows, in, dst, row0);
  };
  auto map = [&](int l) { return tape ? tape + (long long)l * tape_stride : c->cg_fm[g.resnet ? (l & 1 ? 1 : 0) : (l & 1)]; };
  if (tape && n_rows > c->cg_rows) return fail(c, VMC_ERR_STATE, "taped forward beyond one block");
  for (long long blk0 = 0; blk0 < n_rows; blk0 += c->cg_rows) {
    const long long row0 = first_row + blk0;
    const int rows = (int)(n_rows - blk0 < c->cg_rows ? n_rows - blk0 : c->cg_rows);
    const float* last;
    PROPAGATE(conv(0, rows, nullptr, map(0), row0));
    if (!g.resnet) {           // Conv2DNetwork (wavefunctions.py:572-575): act between the convolutions, none behind the last
      for (int l = 1; l < g.n_conv; ++l)
        PROPAGATE(conv(l, rows, map(l - 1), map(l), row0));
      last = map(g.n_conv - 1);
    } else {                   // ResNet2D (wavefunctions.py:766-772; layers.py:226-228): h += second(selu(first(h)))
      for (int l = 1; l + 1 < g.n_conv; l += 2) {
        const float* h = tape ? map(l - 1) : c->cg_fm[0];
        float* u = tape ? map(l) : c->cg_fm[1];
        float* hn = tape ? map(l + 1) : c->cg_fm[0];
        PROPAGATE(conv(l, rows, h, u, row0));
        if (tape) HIPCHK(c, hipMemcpyAsync(hn, h, (size_t)rows * g.N * Fp * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        PROPAGATE(conv(l + 1, rows, u, hn, row0));
      }
      last = tape ? map(g.n_conv - 1) : c->cg_fm[0];
    }
    if (!out) continue;        // (taped forward of the SR matvec: the maps are all that is wanted)
    HIPCHK(c, launch_cgen_rowsum(c->stream, last, rows, g.N, g.F, Fp, c->cg_sum));
    const WideOnsite on{nullptr, nullptr, nullptr, nullptr, nullptr};
    HIPCHK(c, launch_wide_out_part(c->stream, c->cg_sum, 1, c->cg_zero, rows, rowinfo ? rowinfo : c->rowinfo_id, row0,
                                   c->half_jx, p.logit, c->oact, ratio, out, on));
  }
  return VMC_OK;
}

// ---- gradient machinery of the general path, one block of `rows` chains (first chain `row0` of `configs`) at a time
// buffers of the first gradient call
static int cgen_grad_buffers(vmc_ctx* c) {
  const ConvGeom& g = c->cg;
  if (c->cg_tape) return VMC_OK;
  const int T = g.K * g.KW, n_conv = g.n_conv;
  const long long map_floats = c->cg_rows * g.N * cgen_fp(g);
  const int kmax = T * (n_conv > 1 ? g.F : 1) + 1;                      // rows of the largest weight-gradient product
  HIPCHK(c, dalloc(&c->cg_tape, (long long)n_conv * map_floats));
  HIPCHK(c, dalloc(&c->cg_gl, (long long)n_conv * map_floats));
  HIPCHK(c, dalloc(&c->cg_g[0], map_floats));
  HIPCHK(c, dalloc(&c->cg_wpos, c->cg_rows * g.N));
  if (n_conv > 1) HIPCHK(c, dalloc(&c->cg_wt, cgen_off_wt(g, n_conv)));
  c->cg_ws_floats = (long long)CGEN_SPLITK * 2 * kmax * g.F;
  HIPCHK(c, dalloc(&c->cg_ws, c->cg_ws_floats));
  return VMC_OK;
}
static float* cgen_tape(vmc_ctx* c, int l) { return c->cg_tape + (long long)l * c->cg_rows * c->cg.N * cgen_fp(c->cg); }
static float* cgen_gl(vmc_ctx* c, int l) { return c->cg_gl + (long long)l * c->cg_rows * c->cg.N * cgen_fp(c->cg); }

// the input of convolution l gathered into cg_A, as its forward did (from the tape of this block)
static int cgen_gather_input(vmc_ctx* c, int l, int rows, long long row0, const float* configs) {
  const ConvGeom& g = c->cg;
  CgenIm2colArgs a;
  memset(&a, 0, sizeof(a));
  a.g = g; a.layer = l; a.Fp = cgen_fp(g); a.rows = rows; a.lda = plan_cgen_lda(g); a.A = c->cg_A; a.pre_act = -1;
  if (l == 0) { a.src = configs; a.row0 = row0; a.bonds = c->bonds ? c->bonds : c->bond_dummy; }
  else { a.src = cgen_tape(c, l - 1); a.pre_act = cgen_in_pre(c, l); }          // (residual blocks: h / the stored selu(u))
  HIPCHK(c, launch_cgen_im2col(c->stream, a));
  return VMC_OK;
}

// dst (+)= the transposed convolution l (>= 1) of G: the inverse gather against the transposed weight image
static int cgen_input_grad(vmc_ctx* c, int l, int rows, const float* G, float* dst, bool accumulate) {
  const ConvGeom& g = c->cg;
  CgenIm2colArgs a;
  memset(&a, 0, sizeof(a));
  a.g = g; a.layer = l; a.Fp = cgen_fp(g); a.rows = rows; a.lda = plan_cgen_lda(g); a.A = c->cg_A; a.pre_act = -1;
  a.inverse = 1; a.src = G;
  HIPCHK(c, launch_cgen_im2col(c->stream, a));
  GemmArgs m; memset(&m, 0, sizeof(m));
  m.A = c->cg_A; m.sam = a.lda; m.sak = 1;
  m.B = c->cg_wt + cgen_off_wt(g, l); m.sbk = g.F; m.sbn = 1;
  m.M = rows * g.N; m.N = g.F; m.K = g.K * g.KW * g.F; m.C = dst; m.ldc = a.Fp;
  m.epilogue = accumulate ? 3 : 0; m.splitk = 1;
  HIPCHK(c, launch_gemm(c->stream, m));
  return VMC_OK;
}

// cg_gl[l] = d logit / d z_l for every convolution of the block (the tape of the block in cg_tape):
//   G_{l-1} = (transposed convolution l of G_l) (.) f'(z_{l-1}); residual blocks accumulate both branches into d / d h
static int cgen_backward(vmc_ctx* c, int rows, long long row0, const float* oscale) {
  const ConvGeom& g = c->cg;
  const int Fp = cgen_fp(g), n_conv = g.n_conv;
  const long long M = (long long)rows * g.N;
  float* D = c->cg_g[0];
  HIPCHK(c, launch_cgen_fill(c->stream, cgen_gl(c, n_conv - 1), oscale, row0, rows, g.N, g.F, Fp));
  if (!g.resnet) {
    for (int l = n_conv - 1; l >= 1; --l) {
      PROPAGATE(cgen_input_grad(c, l, rows, cgen_gl(c, l), D, false));
      HIPCHK(c, launch_cgen_dact(c->stream, D, cgen_tape(c, l - 1), g.hact, cgen_post(c), M * Fp, g.F, Fp, cgen_gl(c, l - 1)));
    }
  } else {                     // gl[even l] = d / d h behind block (l / 2): the gradient of the block's second convolution
    for (int l2 = n_conv - 1; l2 >= 2; l2 -= 2) {
      const int l1 = l2 - 1;
      PROPAGATE(cgen_input_grad(c, l2, rows, cgen_gl(c, l2), D, false));                                   // d / d selu(u)
      HIPCHK(c, launch_cgen_dact(c->stream, D, cgen_tape(c, l1), CGEN_PRE_SELU, true, M * Fp, g.F, Fp, cgen_gl(c, l1)));   // d / d u (from the stored selu(u))
      HIPCHK(c, hipMemcpyAsync(cgen_gl(c, l2 - 2), cgen_gl(c, l2), (size_t)M * Fp * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
      PROPAGATE(cgen_input_grad(c, l1, rows, cgen_gl(c, l1), cgen_gl(c, l2 - 2), true));                   // d / d h += through the block
    }
  }
  return VMC_OK;
}

// [w_l ; b_l] sums of convolution l: C1 += [im2col(x_l) | 1]^T G (C1 != nullptr), C2 += [im2col(x_l) | 1]^T (kscale (.) G);
// kscale = per-position weights (cg_wpos).  ONE product (k_gemm: dual, implicit ones row, split-K over the positions)
static int cgen_weight_sums(vmc_ctx* c, int l, int rows, long long row0, const float* configs, const float* G,
                            float* C1, float* C2) {
  const ConvGeom& g = c->cg;
  PROPAGATE(cgen_gather_input(c, l, rows, row0, configs));
  const long long M = (long long)rows * g.N;
  GemmArgs m; memset(&m, 0, sizeof(m));
  m.A = c->cg_A; m.sam = 1; m.sak = plan_cgen_lda(g);                       // A(i, k = position) = im2col[k][i]
  m.B = G; m.sbk = cgen_fp(g); m.sbn = 1;
  m.kscale = c->cg_wpos; m.ones_row = 1;
  m.M = cgen_kdim(g, l) + 1; m.N = g.F; m.K = (int)M; m.ldc = g.F;
  if (C1) { m.dual = 1; m.C = C1 + cgen_off_w(g, l); m.C2 = C2 + cgen_off_w(g, l); }
  else { m.dual = 0; m.C = C2 + cgen_off_w(g, l); }                         // the scaled product alone
  m.epilogue = 3; m.splitk = M >= 4096 ? CGEN_SPLITK : 1; m.workspace = c->cg_ws;
  HIPCHK(c, launch_gemm(c->stream, m));
  return VMC_OK;
}

// Gradient sums of the general path: g1 += sum_b O_b, g2 += sum_b w_b O_b (training.py:545-547), a block of chains at a
// time: taped forward (the map of every convolution), d logit / d z_l of every convolution, then
// d / d W_l = im2col(x_l)^T G_l with the bias as an implicit row of ones, one product per convolution for both sums.
static int cgen_gradient_sums(vmc_ctx* c, const float* w) {
  const ConvGeom& g = c->cg;
  ParamSet& p = c->ps[0];
  PROPAGATE(cgen_grad_buffers(c));
  const long long map_floats = c->cg_rows * g.N * cgen_fp(g);
  for (int l = 1; l < g.n_conv; ++l)
    HIPCHK(c, launch_cgen_pack_t(c->stream, p.theta + cgen_off_w(g, l), g.K * g.KW, g.F, c->cg_wt + cgen_off_wt(g, l)));
  if (c->oact != VMC_ACT_EXP_) {
    PROPAGATE(ensure_cache(c, VMC_PSI));
    HIPCHK(c, launch_out_scale(c->stream, p.logit, c->oscale, c->B, c->oact));
  }
  for (long long row0 = 0; row0 < c->B; row0 += c->cg_rows) {
    const int rows = (int)(c->B - row0 < c->cg_rows ? c->B - row0 : c->cg_rows);
    PROPAGATE(cgen_forward(c, VMC_PSI, c->configs, nullptr, rows, nullptr, nullptr, false, c->cg_lnew, c->cg_tape,
                           map_floats, row0));     // (its logits land in cg_lnew[row0 ..]: unused)
    HIPCHK(c, launch_cgen_wpos(c->stream, w, row0, rows, g.N, c->cg_wpos));
    PROPAGATE(cgen_backward(c, rows, row0, c->oact != VMC_ACT_EXP_ ? c->oscale : nullptr));
    for (int l = g.n_conv - 1; l >= 0; --l)
      PROPAGATE(cgen_weight_sums(c, l, rows, row0, c->configs, cgen_gl(c, l), c->acc, c->acc + c->P));
  }
  return VMC_OK;
}

// SR matvec of the general path over the `n_rows` stored chains (sr_cfg): u = sum_b (t_b - c) O_b with t_b = O_b . v,
// u[P] = sum_b (t_b - c).  k_sr_q forms q = u / n - <O> u[P] / n + lambda p, which is S v + lambda v for ANY constant c
// subtracted from every t_b -- and with c = the mean of t the cancellation of <O (O.v)> - <O><O.v> happens per sample,
// before the fp32 sums (the uncentred form of this matvec met the 5e-4 bound on S v but its solutions missed the 1 %
// bound on O_c x).  The constant must be the same on every rank: a sharded solve all-reduces sum_b t_b between the two
// phases (sr_solve_impl), a single rank takes its own mean.  Only the chains are stored: every CG iteration re-runs the
// taped forward and the backward of a block; t_b = sum_l < G_l , im2col(x_l) V_l + v_l > is one more product per
// convolution against the slice of v (phase 1), then the weight sums with t_b - c as the k-scale (phase 2; several
// blocks: a second forward / backward pass, the mean needs every t first).
static int cgen_sr_fwd_bwd(vmc_ctx* c, long long row0, int rows) {
  const long long map_floats = c->cg_rows * c->cg.N * cgen_fp(c->cg);
  PROPAGATE(cgen_forward(c, VMC_PSI, c->sr_cfg, nullptr, rows, nullptr, nullptr, false, nullptr, c->cg_tape, map_floats, row0));
  return cgen_backward(c, rows, row0, nullptr);
}
static int cgen_sr_phase1(vmc_ctx* c, const float* v, int n_rows) {        // sr_t[b] = O_b . v
  const ConvGeom& g = c->cg;
  ParamSet& p = c->ps[0];
  PROPAGATE(cgen_grad_buffers(c));
  if (!c->cg_td) { HIPCHK(c, dalloc(&c->cg_td, c->cg_rows)); HIPCHK(c, dalloc(&c->cg_centre, 1)); }
  const int Fp = cgen_fp(g), lda = plan_cgen_lda(g);
  for (int l = 1; l < g.n_conv; ++l)
    HIPCHK(c, launch_cgen_pack_t(c->stream, p.theta + cgen_off_w(g, l), g.K * g.KW, g.F, c->cg_wt + cgen_off_wt(g, l)));
  for (long long row0 = 0; row0 < n_rows; row0 += c->cg_rows) {
    const int rows = (int)(n_rows - row0 < c->cg_rows ? n_rows - row0 : c->cg_rows);
    PROPAGATE(cgen_sr_fwd_bwd(c, row0, rows));
    for (int l = 0; l < g.n_conv; ++l) {
      PROPAGATE(cgen_gather_input(c, l, rows, row0, c->sr_cfg));
      GemmArgs m; memset(&m, 0, sizeof(m));
      m.A = c->cg_A; m.sam = lda; m.sak = 1;
      m.B = v + cgen_off_w(g, l); m.sbk = g.F; m.sbn = 1;
      m.M = rows * g.N; m.N = g.F; m.K = cgen_kdim(g, l); m.C = c->cg_g[0]; m.ldc = Fp;
      m.bias = v + cgen_off_b(g, l); m.epilogue = 4; m.splitk = 1;
      HIPCHK(c, launch_gemm(c->stream, m));
      HIPCHK(c, launch_cgen_pairdot(c->stream, c->cg_g[0], cgen_gl(c, l), rows, g.N, g.F, Fp, c->cg_td, l == 0));
    }
    HIPCHK(c, launch_cgen_tstore(c->stream, c->cg_td, rows, c->sr_t + row0));
  }
  return VMC_OK;
}
static int cgen_sr_phase2(vmc_ctx* c, int n_rows) {                         // sr_u[0 .. P) += sum_b (t_b - *cg_centre) O_b
  const ConvGeom& g = c->cg;
  const bool one_block = n_rows <= c->cg_rows;       // (then the tapes and G_l of phase 1 are still in place)
  for (long long row0 = 0; row0 < n_rows; row0 += c->cg_rows) {
    const int rows = (int)(n_rows - row0 < c->cg_rows ? n_rows - row0 : c->cg_rows);
    if (!one_block) PROPAGATE(cgen_sr_fwd_bwd(c, row0, rows));
    HIPCHK(c, launch_cgen_wpos_centred(c->stream, c->sr_t, c->cg_centre, row0, rows, g.N, c->cg_wpos));
    for (int l = g.n_conv - 1; l >= 0; --l)
      PROPAGATE(cgen_weight_sums(c, l, rows, row0, c->sr_cfg, cgen_gl(c, l), nullptr, c->sr_u));
  }
  return VMC_OK;
}
// one rank
static int cgen_sr_matvec(vmc_ctx* c, const float* v, int n_rows) {
  PROPAGATE(cgen_sr_phase1(c, v, n_rows));
  HIPCHK(c, launch_cgen_tmean(c->stream, c->sr_t, n_rows, c->cg_centre, c->sr_u + c->P));
  return cgen_sr_phase2(c, n_rows);
}

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
static int wide_stage_act(const vmc_ctx* c, int l) {
  return (c->rbm && l == c->n_hh) ? VMC_ACT_LOGCOSH_ : c->hact;
}

// The last H x H layer of a forward whose activations nobody reads: turn its GEMM into the row-dot form (the output
// layer's dot product as column-tile partials in c->wide_dot, nothing stored to C) where a tile kernel takes the
// shape.  CGS_VMC_ROWDOT=0: never (A/B measurements, tests).  Returns whether `g` was changed.
static bool wide_rowdot(vmc_ctx* c, const ParamSet& p, GemmArgs& g) {
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
int local_energy_device(vmc_ctx* c, int which, bool defer_reduce = false, bool* deferred = nullptr) {
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

// ---------------------------------------------------------------- collectives
// RCCL is resolved at first use with dlopen -- the copy already loaded into the process (torch's)
// if there is one -- so the library itself carries no link-time dependency on it.
struct RcclUniqueId { char internal[128]; };   // ncclUniqueId
struct Rccl {
  int (*all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*get_unique_id)(RcclUniqueId*) = nullptr;
  int (*comm_init_rank)(void**, int, RcclUniqueId, int) = nullptr;
  int (*comm_destroy)(void*) = nullptr;
  const char* (*get_error_string)(int) = nullptr;
};
std::string g_rccl_error;

const Rccl* rccl() {
  static Rccl r;
  static bool tried = false, ok = false;
  if (!tried) {
    tried = true;
    // The librccl that belongs to the HIP runtime THIS library is bound to: a process may hold two ROCm
    // stacks (torch bundles libamdhip64 / librccl next to the system's, same sonames), and a
    // communicator of the other stack's librccl would launch through the other runtime on this one's
    // streams and buffers.  So: the directory of the libamdhip64 behind our hip* symbols first.
    void* h = nullptr;
    std::vector<std::string> names;
    Dl_info info;
    if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {
      std::string dir(info.dli_fname);
      const size_t slash = dir.rfind('/');
      if (slash != std::string::npos) {
        dir.resize(slash);
        names.push_back(dir + "/librccl.so.1");
        names.push_back(dir + "/librccl.so");
      }
    }
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    for (const std::string& name : names)
      if (!h) h = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (h) {
      r.all_reduce = (decltype(r.all_reduce))dlsym(h, "ncclAllReduce");
      r.get_unique_id = (decltype(r.get_unique_id))dlsym(h, "ncclGetUniqueId");
      r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(h, "ncclCommInitRank");
      r.comm_destroy = (decltype(r.comm_destroy))dlsym(h, "ncclCommDestroy");
      r.get_error_string = (decltype(r.get_error_string))dlsym(h, "ncclGetErrorString");
    }
    ok = r.all_reduce && r.get_unique_id && r.comm_init_rank && r.comm_destroy;
    if (!ok) g_rccl_error = "librccl.so / its nccl* entry points not found";
  }
  return ok ? &r : nullptr;
}

std::string rccl_error_string(const Rccl* r, int rc) {
  return (r && r->get_error_string) ? std::string(r->get_error_string(rc)) : "code " + std::to_string(rc);
}

// In-place all-reduce of n floats at device pointer buf, ordered on the ctx's stream.
//   comm != NULL                : ncclAllReduce on the stream (no host synchronisation)
//   comm == NULL, world <= 1    : nothing to do
//   comm == NULL, world  > 1    : the registered host hook, staged through pinned host memory
//   comm == NULL, world  > 1    : the device hook (the host's collective library reduces the device
//                                 buffer in stream order), else the host hook through pinned memory
// op == VMC_REDUCE_SUM_F64: buf holds n doubles.
int reduce_buffer(vmc_ctx* c, void* comm, int world, void* buf, long long n, int op) {
  const bool f64 = op == VMC_REDUCE_SUM_F64;
  if (comm) {
    const Rccl* r = rccl();
    if (!r) return fail(c, VMC_ERR_UNSUPPORTED, g_rccl_error);
    const int rc = r->all_reduce(buf, buf, (size_t)n, f64 ? /*ncclFloat64*/ 8 : /*ncclFloat32*/ 7,
                                 op == VMC_REDUCE_MAX ? /*ncclMax*/ 2 : /*ncclSum*/ 0, comm, c->stream);
    if (rc != 0) return fail(c, VMC_ERR_HIP, "ncclAllReduce: " + rccl_error_string(r, rc));
    return VMC_OK;
  }
  if (world <= 1) return VMC_OK;
  if (c->dev_reduce) {
    const int rc = c->dev_reduce(c->dev_reduce_user, buf, n, op, (void*)c->stream);
    if (rc != 0) return fail(c, VMC_ERR_HIP, "device all-reduce hook failed with code " + std::to_string(rc));
    return VMC_OK;
  }
  if (!c->host_reduce)
    return fail(c, VMC_ERR_STATE, "world_size > 1 needs an RCCL communicator, vmc_set_device_allreduce or vmc_set_host_allreduce");
  if (f64 && !(c->host_reduce_caps & VMC_HOST_REDUCE_CAP_F64))
    return fail(c, VMC_ERR_UNSUPPORTED, "the registered host all-reduce hook has not declared float64 support "
                "(vmc_set_host_allreduce_caps(ctx, VMC_HOST_REDUCE_CAP_F64)): it would be handed doubles");
  const long long nf = f64 ? 2 * n : n;          // staging size in floats
  if (nf > c->h_stage_n) {
    if (c->h_stage) hipHostFree(c->h_stage);
    c->h_stage = nullptr; c->h_stage_n = 0;
    HIPCHK(c, hipHostMalloc((void**)&c->h_stage, (size_t)nf * sizeof(float), hipHostMallocDefault));
    c->h_stage_n = nf;
  }
  HIPCHK(c, hipMemcpyAsync(c->h_stage, buf, nf * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int rc = c->host_reduce(c->host_reduce_user, c->h_stage, n, op);
  if (rc != 0) return fail(c, VMC_ERR_HIP, "host all-reduce hook failed with code " + std::to_string(rc));
  HIPCHK(c, hipMemcpyAsync(buf, c->h_stage, nf * sizeof(float), hipMemcpyHostToDevice, c->stream));
  return VMC_OK;
}

bool sharded(void* comm, int world) { return comm != nullptr || world > 1; }

int reduce_accumulators(vmc_ctx* c, void* comm, int world) {
  if (!sharded(comm, world)) return VMC_OK;
  PROPAGATE(acc_zeros(c));
  PROPAGATE(reduce_buffer(c, comm, world, c->acc, 2 * c->P + 8, VMC_REDUCE_SUM));
  HIPCHK(c, launch_scale_one(c->stream, c->acc + 2 * c->P + 4, 1.f / (float)(world > 1 ? world : 1)));
  return VMC_OK;
}

}  // namespace

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
    const char* fg = getenv("CGS_VMC_CONV_GENERAL");      // =1: the general convolution path for every shape (tests, A/B runs)
    const int rc = plan_desc(d, !(wf && atoi(wf) == 0), &dp, msg, sizeof(msg), fg && atoi(fg) != 0);
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
    CA(dalloc(&c->cg_A, rows * cg.N * plan_cgen_lda(cg)));
    for (int i = 0; i < 2; ++i) CA(dalloc(&c->cg_fm[i], rows * cg.N * cgen_fp(cg)));
    CA(dalloc(&c->cg_sum, rows)); CA(dalloc(&c->cg_zero, 1)); CA(dalloc(&c->cg_lnew, B));
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
  for (float* q : {c->cg_A, c->cg_fm[0], c->cg_fm[1], c->cg_zero, c->cg_lnew, c->cg_tape, c->cg_gl, c->cg_g[0], c->cg_g[1], c->cg_wpos,
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
static bool sampler_refresh_ok(const vmc_ctx* c) {
  static const bool on = !(getenv("CGS_VMC_SAMPLER_REFRESH") && atoi(getenv("CGS_VMC_SAMPLER_REFRESH")) == 0);
  return on && !c->conv && !(c->wide && !c->wide_fast);
}
static int refresh_cache_by_sampler(vmc_ctx* c, int which) {
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

int vmc_accumulators_devptr(vmc_ctx* c, void** dev_ptr, int64_t* n_floats) {
  CHECK_CTX(c);
  PROPAGATE(acc_zeros(c));
  if (dev_ptr) *dev_ptr = c->acc;
  if (n_floats) *n_floats = 2 * c->P + 8;
  return VMC_OK;
}

// In-place SUM all-reduce of the accumulator buffer, stream-ordered on the ctx's stream (transport:
// see reduce_buffer).  g_count (number of accumulate calls, identical on every rank) is divided
// back by the world size so that sharded and unsharded gradients agree (cgs_vmc_amd/parallel.py).
int vmc_allreduce_accumulators(vmc_ctx* c, void* nccl_comm, int32_t world_size) {
  CHECK_CTX(c);
  PROPAGATE(acc_zeros(c));
  PROPAGATE(reduce_accumulators(c, nccl_comm, world_size));
  return VMC_OK;
}

int vmc_set_host_allreduce(vmc_ctx* c, vmc_host_allreduce_fn hook, void* user) {
  CHECK_CTX(c);
  c->host_reduce = hook;
  c->host_reduce_user = user;
  c->host_reduce_caps = 0;          // a new hook has declared nothing yet
  return VMC_OK;
}

int vmc_set_host_allreduce_caps(vmc_ctx* c, int32_t caps) {
  CHECK_CTX(c);
  if (caps & ~VMC_HOST_REDUCE_CAP_F64) return fail(c, VMC_ERR_INVALID, "unknown capability bits");
  c->host_reduce_caps = caps;
  return VMC_OK;
}

int vmc_set_device_allreduce(vmc_ctx* c, vmc_device_allreduce_fn hook, void* user) {
  CHECK_CTX(c);
  c->dev_reduce = hook;
  c->dev_reduce_user = user;
  return VMC_OK;
}

const char* vmc_rccl_last_error(void) { return g_rccl_error.c_str(); }

const char* vmc_rccl_library_path(void) {
  static std::string path;
  const Rccl* r = rccl();
  Dl_info info;
  if (r && dladdr((void*)r->all_reduce, &info) && info.dli_fname) path = info.dli_fname;
  return path.c_str();
}

int vmc_device_pci_bus_id(int32_t device, char* buf, int32_t len) {
  if (!buf || len < 16) return VMC_ERR_INVALID;
  buf[0] = 0;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return VMC_ERR_HIP;
  if (device < 0 || device >= n) return VMC_ERR_INVALID;
  return hipDeviceGetPCIBusId(buf, len, device) == hipSuccess ? VMC_OK : VMC_ERR_HIP;
}

const char* vmc_hip_runtime_path(void) {
  static std::string path;
  Dl_info info;
  if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) path = info.dli_fname;
  return path.c_str();
}

int vmc_rccl_unique_id(uint8_t id[128]) {
  if (!id) { g_rccl_error = "null id"; return VMC_ERR_INVALID; }
  const Rccl* r = rccl();
  if (!r) return VMC_ERR_UNSUPPORTED;
  RcclUniqueId u;
  const int rc = r->get_unique_id(&u);
  if (rc != 0) { g_rccl_error = std::string("ncclGetUniqueId: ") + rccl_error_string(r, rc); return VMC_ERR_HIP; }
  memcpy(id, u.internal, sizeof(u.internal));
  return VMC_OK;
}

int vmc_rccl_comm_create(const uint8_t id[128], int32_t world_size, int32_t rank, int32_t device,
                         void** nccl_comm) {
  if (!id || !nccl_comm || world_size < 1 || rank < 0 || rank >= world_size) {
    g_rccl_error = "bad communicator arguments";
    return VMC_ERR_INVALID;
  }
  *nccl_comm = nullptr;
  const Rccl* r = rccl();
  if (!r) return VMC_ERR_UNSUPPORTED;
  DeviceGuard guard(device);
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != device) { g_rccl_error = "cannot select the device"; return VMC_ERR_HIP; }
  RcclUniqueId u;
  memcpy(u.internal, id, sizeof(u.internal));
  const int rc = r->comm_init_rank(nccl_comm, world_size, u, rank);
  if (rc != 0) { g_rccl_error = std::string("ncclCommInitRank: ") + rccl_error_string(r, rc); *nccl_comm = nullptr; return VMC_ERR_HIP; }
  return VMC_OK;
}

int vmc_rccl_comm_destroy(void* nccl_comm) {
  if (!nccl_comm) return VMC_OK;
  const Rccl* r = rccl();
  if (!r) return VMC_ERR_UNSUPPORTED;
  const int rc = r->comm_destroy(nccl_comm);
  if (rc != 0) { g_rccl_error = std::string("ncclCommDestroy: ") + rccl_error_string(r, rc); return VMC_ERR_HIP; }
  return VMC_OK;
}

int vmc_debug_allreduce(vmc_ctx* c, void* nccl_comm, int32_t world_size, float* host, int64_t n, int32_t op) {
  ENTER(c);
  if (!host || n < 1 || (op != VMC_REDUCE_SUM && op != VMC_REDUCE_MAX)) return fail(c, VMC_ERR_INVALID, "bad arguments");
  if (n > c->d_stage_n) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->d_stage) hipFree(c->d_stage);
    c->d_stage = nullptr; c->d_stage_n = 0;
    HIPCHK(c, dalloc(&c->d_stage, n));
    c->d_stage_n = n;
  }
  HIPCHK(c, hipMemcpyAsync(c->d_stage, host, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
  PROPAGATE(reduce_buffer(c, nccl_comm, world_size, c->d_stage, n, op));
  HIPCHK(c, hipMemcpyAsync(host, c->d_stage, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
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
