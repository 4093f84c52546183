// Internal header of the host side of libcgsvmc_hip.so (round 6: vmc_api.hip split by kernel path behind the same
// include/cgsvmc.h): the ctx, the entry-point macros, the small inline helpers and the prototypes of what the
// translation units vmc_api*.hip share.  Not installed; nothing outside csrc/ includes it.
#pragma once
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

namespace vmcapi {


extern std::string g_create_error;   // vmc_api.hip

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

}  // namespace vmcapi
using vmcapi::ParamSet;
using vmcapi::TimedRegion;

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
  long long cg_rows = 0;                   // row configurations per block (sized by the im2col matrix: the GEMM form, the gradient path)
  long long cg_rows_fwd = 0;               // ... of an untaped forward whose convolutions all run on the band kernel (sized by the two maps)
  float* cg_A = nullptr;                   // im2col rows [cg_rows * N][plan_cgen_lda]
  float* cg_fm[2] = {nullptr, nullptr};    // feature maps [cg_rows][N][Fp] (cgen_post: activations; the cosine: pre-activations)
  double* cg_sum = nullptr;                // [cg_rows] sums of the last map
  float* cg_zero = nullptr;                // one 0.f (the "b_out" of wide_out_finish)
  float* cg_lnew = nullptr;                // [B] candidate logits of the sampler
  // the sampler's chain groups (run_sweep_cgen): group 0 on `stream`, the others on streams of their own, so that the
  // partly filled last round of one group's launch runs beside the next launch of another
  hipStream_t cg_grp_stream[3] = {nullptr, nullptr, nullptr};
  hipEvent_t cg_grp_ev[4] = {nullptr, nullptr, nullptr, nullptr};    // [0]: `stream` is ready; [g]: group g has finished
  hipStream_t cg_stream_cur = nullptr;     // the stream cgen_conv / cgen_forward launch on (null: `stream`)
  long long cg_map_row0 = 0;               // first row of cg_fm / cg_A an untaped forward writes (a group's slice)
  float* cg_pmaps = nullptr;               // [n_conv][B][N][Fp] the chains' maps of every convolution (the patch sampler, conv_patch.hip)
  // ... its gradient path (allocated by the first gradient call): the map of every convolution (the tape), two
  // d logit / d map buffers, per-position weights, the transposed weight images, the split-K workspace
  float* cg_tape = nullptr; float* cg_gl = nullptr; float* cg_g[2] = {nullptr, nullptr}; float* cg_wpos = nullptr; float* cg_wt = nullptr;
  float* cg_ws = nullptr; long long cg_ws_floats = 0;
  double* cg_td = nullptr;                 // [cg_rows] O_b . v of a block (SR)
  long long cg_sr_tape_rows = 0;           // > 0: cg_tape / cg_gl hold the taped forward and the backward of the first that many STORED chains
                                           // at the parameters of the running solve (one block: kept across its CG iterations)
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


namespace vmcapi {


inline int fail(vmc_ctx* c, int code, const std::string& msg) {
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
inline int sweep_cus(const vmc_ctx* c) { return c->sweep_tile == 8 ? (c->B + 7) / 8 : (c->B + 15) / 16; }

// The sampler may overtake the accumulate enqueued just before it when it leaves the local-energy
// kernel at least a quarter of the CUs; with one 16-chain tile per CU (config 3) there is nothing
// to share and the launch stays on `stream`.
inline bool can_overlap(const vmc_ctx* c) {
  return c->overlap && (c->overlap_full || sweep_cus(c) <= (3 * c->num_cus) / 4);
}

// `acc` is about to be read or partially written: turn a pending reset into real zeros
inline int acc_zeros(vmc_ctx* c) {
  if (c->acc_fresh) {
    hipError_t e = hipMemsetAsync(c->acc, 0, (2 * c->P + 8) * sizeof(float), c->stream);
    if (e != hipSuccess) return fail(c, VMC_ERR_HIP, std::string("hipMemsetAsync: ") + hipGetErrorString(e));
    c->acc_fresh = false;
  }
  return VMC_OK;
}

inline int join_sweep(vmc_ctx* c) {
  if (c->sweep_pending) {
    hipError_t e = hipStreamWaitEvent(c->stream, c->ev_sweep_done, 0);
    if (e != hipSuccess) return fail(c, VMC_ERR_HIP, std::string("hipStreamWaitEvent: ") + hipGetErrorString(e));
    c->sweep_pending = false;
  }
  return VMC_OK;
}

inline void swap_chain_buffers(vmc_ctx* c) {
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
inline void account(vmc_ctx* c, const TimedRegion& r) {
  float ms = 0.f;
  hipEventElapsedTime(&ms, r.start, r.stop);
  auto& t = c->timings[r.name];
  t.first += ms; t.second += 1;
  c->event_pool.emplace_back(r.start, r.stop);
}

inline void harvest_finished(vmc_ctx* c) {
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

inline void drain_timings(vmc_ctx* c) {
  for (auto& r : c->pending) {
    hipEventSynchronize(r.stop);
    account(c, r);
  }
  c->pending.clear();
}

inline long long off_w(const vmc_ctx* c, int l) { return plan_off_w(c->lay, c->H, l); }   // weight matrix of layer l (0 = first)
inline long long off_b(const vmc_ctx* c, int l) { return plan_off_b(c->lay, c->H, l); }   // biases sit right behind their weights
inline long long off_wout(const vmc_ctx* c) { return c->lay.off_wout; }
inline long long off_bout(const vmc_ctx* c) { return c->lay.off_bout; }


// ---- shared across the translation units (definitions: the file named)
// vmc_api.hip
int ensure_packed(vmc_ctx* c, int which);
hipError_t launch_rows(vmc_ctx* c, int which, const TailArgs& a, bool ratio);
TailArgs tail_args(vmc_ctx* c, int which);
ConvParams conv_params(const ParamSet& p);
int ensure_cache(vmc_ctx* c, int which);
int conv_rows(vmc_ctx* c, int which, const float* configs, const int2* rowinfo, int rows,
              const int* rows_dev, bool ratio, float* out, bool with_tape);
int first_layer(vmc_ctx* c, const ParamSet& p, const float* configs, float* z1, int rows);
int wide_stage_act(const vmc_ctx* c, int l);
bool wide_rowdot(vmc_ctx* c, const ParamSet& p, GemmArgs& g);
int wide_forward(vmc_ctx* c, int which, const float* z1, const int2* rowinfo, long long n_rows, bool ratio,
                 float* out, const float* onsite);
void invalidate_configs(vmc_ctx* c);
int ensure_list(vmc_ctx* c);
int local_energy_device(vmc_ctx* c, int which, bool defer_reduce = false, bool* deferred = nullptr);
int grow_tmp(vmc_ctx* c, long long rows);
// vmc_api_cgen.hip (the general convolution path)
int cgen_forward(vmc_ctx* c, int which, const float* configs, const int2* rowinfo, long long n_rows,
                 const int* iup, const int* idn, bool ratio, float* out, float* tape = nullptr,
                 long long tape_stride = 0, long long first_row = 0);
int cgen_patch_mode(const vmc_ctx* c);
int cgen_patch_maps(vmc_ctx* c, int which);
void cgen_patch_args(const vmc_ctx* c, int which, CgenPatchArgs* a);
bool cgen_single_block(const vmc_ctx* c, long long n_rows);
const float* cgen_last_map(const vmc_ctx* c);
int cgen_gradient_sums(vmc_ctx* c, const float* w);
int cgen_sr_phase1(vmc_ctx* c, const float* v, int n_rows);
int cgen_sr_phase2(vmc_ctx* c, int n_rows);
int cgen_sr_matvec(vmc_ctx* c, const float* v, int n_rows);
// vmc_api_coll.hip
int reduce_buffer(vmc_ctx* c, void* comm, int world, void* buf, long long n, int op);
bool sharded(void* comm, int world);
int reduce_accumulators(vmc_ctx* c, void* comm, int world);
// vmc_api_sweep.hip
bool sampler_refresh_ok(const vmc_ctx* c);
int refresh_cache_by_sampler(vmc_ctx* c, int which);
}  // namespace vmcapi
