// Shared device helpers and launcher declarations for libcgsvmc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plan.hpp"   // every pure-host decision (layouts, LDS budgets, grids); no HIP in it

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VMC_WAVE 64

// --------------------------------------------------------------------------------------
// Philox4x32-10 (counter-based RNG).  Counter = (block, global chain id, step_lo, step_hi),
// key = (seed_lo, seed_hi); block b gives the site uniforms 4b..4b+3 of one mc_step
// (graph_builders.py:59), block 0xFFFFFFFF word 0 the acceptance uniform (76-77).
// --------------------------------------------------------------------------------------
// one Philox round (+ key bump); ten of them make Philox4x32-10
__device__ __forceinline__ void philox_round(uint4& c, uint2& k) {
  const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
  const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
  c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
  k.x += 0x9E3779B9u;
  k.y += 0xBB67AE85u;
}

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
  for (int r = 0; r < 10; ++r) philox_round(c, k);
  return c;
}

// The same ten rounds with each 32 x 32 -> 64 bit product as ONE v_mad_u64_u32 (the form above compiles to
// v_mul_hi_u32 + v_mul_lo_u32, both quarter rate): 272 against 356 clocks per un-overlapped call
// (tools/ubench/philox_rate.hip).  For callers that draw OUTSIDE an MFMA phase (k_sweep8); k_sweep16's Philox pieces
// between MFMA groups were tuned around the form above and ran ~1 % slower with this one.
__device__ __forceinline__ uint4 philox4x32_10_wide(uint4 c, uint2 k) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c.x, p1 = (unsigned long long)0xCD9E8D57u * c.z;
    c = make_uint4((uint32_t)(p1 >> 32) ^ c.y ^ k.x, (uint32_t)p1, (uint32_t)(p0 >> 32) ^ c.w ^ k.y, (uint32_t)p0);
    k.x += 0x9E3779B9u;
    k.y += 0xBB67AE85u;
  }
  return c;
}

__device__ __forceinline__ float u32_to_uniform(uint32_t x) {
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}

#define VMC_ACCEPT_BLOCK 0xFFFFFFFFu

// --------------------------------------------------------------------------------------
// Device-side view of one parameter set, re-packed for the kernels (pack.hip).
//   Hp       layer_size rounded up to a multiple of 64 (zero padded)
//   w1p      [N][Hp]      first layer, row n = site n
//   b1p      [Hp]
//   bh       [L-1][Hp]    biases of the H x H layers
//   p16      [L-1][Hp/16 (to)][Hp/16 (ti)][64 (lane)][4 (e)]
//              = W[16ti + 4(lane>>4) + e][16to + (lane&15)]        (A operand, 16x16x4)
//   p16t     same shape, = W[16to + (lane&15)][16ti + 4(lane>>4) + e]: the transposed image of
//            the back-propagation chain (k_backprop16)
//   woutp    [Hp], bout [1]
// RBM ansatz (RestrictedBoltzmannNetwork, wavefunctions.py:391-452): same image with one more
// H x H layer whose output goes through log cosh instead of relu; woutp = ones, bout = the
// onsite bias, won [N] = the onsite weights (x . w_on is added to the logit).
// --------------------------------------------------------------------------------------
struct PackedParams {
  const float* w1p;
  const float* b1p;
  const float* bh;
  const float* p16;
  const float* woutp;
  const float* bout;
  const float* won;   // [N] onsite weights (RBM) or nullptr
};

// (ParamLayout, the offsets of the pieces of the flat parameter vector: plan.hpp)

// log cosh(z) = |z| + log(1 + exp(-2|z|)) - log 2 (finite for every z, unlike log(cosh(z)))
__device__ __forceinline__ float vmc_logcosh(float z) {
  const float a = fabsf(z);
  // same expression in every kernel; hardware exp / log: the argument of the log is in (1, 2],
  // so its absolute error is ~1e-7, below the rounding of the sum over H units
  return a + __logf(1.f + __expf(-2.f * a)) - 0.69314718056f;
}

// --------------------------------------------------------------------------------------
// layers.NONLINEARITIES (layers.py:13-21), ids of include/cgsvmc.h VMC_ACT_*.
// Hidden activations are template parameters of the fused kernels (relu is the tuned path); the
// output activation only touches scalars per row and is a run-time id.
// cos / sin go through the hardware v_cos_f32 / v_sin_f32 after an fp32 range reduction
// (|error| ~ 1e-6 for |z| < ~50, stated with the tolerance of the activation tests); exp / tanh
// use the ocml device functions.
// --------------------------------------------------------------------------------------
#define VMC_ACT_RELU_ 0
#define VMC_ACT_EXP_ 1
#define VMC_ACT_COS_ 2
#define VMC_ACT_TAN_ 3
#define VMC_ACT_TANH_ 4
#define VMC_ACT_SIGMOID_ 5
#define VMC_ACT_IDENTITY_ 6
// run-time only: the RBM's last-stage log cosh as an "activation" of the general (wide.hip) path
#define VMC_ACT_LOGCOSH_ 100

// The relu below is an inline-asm v_max_f32, and the compiler's hazard recogniser does not look inside an asm
// statement: where it reads an MFMA result directly -- no compiler-visible instruction (a copy out of an AGPR, an
// add) in between -- no wait states are inserted and it may see the accumulator before the last MFMAs have written
// it (round 5: the split sampler lost the last k-step of a layer in 4 .. 8 of 16 output rows that way).  The fp32
// samplers had the same shape of code behind v_mfma_f32_16x16x4_f32 chains (tools/check_mfma_read_hazard.py: 200
// sites, up to all 10 wait states of LLVM's table missing) and passed every parity test, i.e. rested on an
// interlock nothing documents; since round 6 they settle too and the checker is strict for every MFMA form.
// vmc_mfma_settle* stand between the last MFMA of a chain and such a reader: EVERY accumulator the reader touches
// passes through the statement (so it follows the MFMAs and precedes the readers; an accumulator left out could
// have its last MFMAs scheduled behind the wait) and it holds the wait states itself -- issued while the last MFMA
// still occupies the matrix pipe.  16 wait states cover every form used here (LLVM's table: 10 for the f32
// 16x16x4 form, 7 for bf16 16x16x32, 4 for f32 4x4x1).
__device__ __forceinline__ void vmc_mfma_settle(f32x4& a, f32x4& b) {
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a), "+v"(b));
}
template <int N>
__device__ __forceinline__ void vmc_mfma_settle_all(f32x4 (&acc)[N]) {
  static_assert(N >= 1 && N <= 4, "one asm operand per accumulator");
  if constexpr (N == 1) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0]));
  if constexpr (N == 2) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]));
  if constexpr (N == 3) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]));
  if constexpr (N == 4) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
}

template <int ACT>
__device__ __forceinline__ float vmc_act(float z) {
  if (ACT == VMC_ACT_RELU_) {
    // one v_max_f32: fmaxf() costs a second one (LLVM quiets a possible signalling NaN first), and
    // VALU instructions next to MFMAs are paid in matrix time; max(0, NaN) = 0 either way
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(z));
    return r;
  }
  if (ACT == VMC_ACT_EXP_) return expf(z);
  if (ACT == VMC_ACT_COS_) return __cosf(z);
  if (ACT == VMC_ACT_TAN_) return __sinf(z) / __cosf(z);
  if (ACT == VMC_ACT_TANH_) return tanhf(z);
  if (ACT == VMC_ACT_SIGMOID_) return 1.f / (1.f + expf(-z));
  return z;
}

// f'(z) from the activation value a = f(z); the cosine needs sin z, which the producers of the
// activations store next to them (`dact` arrays) instead
template <int ACT>
__device__ __forceinline__ float vmc_dact_from_a(float a) {
  if (ACT == VMC_ACT_RELU_) return a > 0.f ? 1.f : 0.f;
  if (ACT == VMC_ACT_EXP_) return a;
  if (ACT == VMC_ACT_TAN_) return 1.f + a * a;
  if (ACT == VMC_ACT_TANH_) return 1.f - a * a;
  if (ACT == VMC_ACT_SIGMOID_) return a * (1.f - a);
  return 1.f;   // identity (cos: see dact arrays)
}

__device__ __forceinline__ float vmc_act_rt(int act, float z) {
  switch (act) {
    case VMC_ACT_RELU_: return vmc_act<VMC_ACT_RELU_>(z);
    case VMC_ACT_EXP_: return vmc_act<VMC_ACT_EXP_>(z);
    case VMC_ACT_COS_: return vmc_act<VMC_ACT_COS_>(z);
    case VMC_ACT_TAN_: return vmc_act<VMC_ACT_TAN_>(z);
    case VMC_ACT_TANH_: return vmc_act<VMC_ACT_TANH_>(z);
    case VMC_ACT_SIGMOID_: return vmc_act<VMC_ACT_SIGMOID_>(z);
    case VMC_ACT_LOGCOSH_: return vmc_logcosh(z);
    default: return z;
  }
}

__device__ __forceinline__ float vmc_dact_rt(int act, float z, float a) {
  switch (act) {
    case VMC_ACT_RELU_: return z > 0.f ? 1.f : 0.f;
    case VMC_ACT_EXP_: return a;
    case VMC_ACT_COS_: return -__sinf(z);
    case VMC_ACT_TAN_: return 1.f + a * a;
    case VMC_ACT_TANH_: return 1.f - a * a;
    case VMC_ACT_SIGMOID_: return a * (1.f - a);
    default: return 1.f;
  }
}

// Output activation g (wavefunctions.py:350-353).  exp: psi = exp(x - shift), every ratio is
// exp(x' - x); any other g: psi = g(x) with no shift and ratios are taken in the linear domain.
//   ratio psi'/psi (signed, local energy: operators.py:259)
__device__ __forceinline__ float vmc_out_ratio(int oact, float x_new, float x_old) {
  if (oact == VMC_ACT_EXP_) return expf(x_new - x_old);
  return vmc_act_rt(oact, x_new) / vmc_act_rt(oact, x_old);
}
//   Metropolis test |psi'| / |psi| > sqrt(u) (graph_builders.py:75-79); half_log_u = 0.5 log u
__device__ __forceinline__ bool vmc_out_accept(int oact, float x_new, float x_old, float u,
                                               float half_log_u) {
  if (oact == VMC_ACT_EXP_) return (x_new - x_old) > half_log_u;
  const float pn = fabsf(vmc_act_rt(oact, x_new)), po = fabsf(vmc_act_rt(oact, x_old));
  return pn / po > sqrtf(u);     // 0/0 = nan > . is false, x/0 = inf accepts: as the reference
}
//   (1/psi) d psi / d x: the per-sample factor of O_k (training.py:545: grad of psi / stop_gradient(psi))
__device__ __forceinline__ float vmc_out_dlog(int oact, float x) {
  if (oact == VMC_ACT_EXP_) return 1.f;
  const float a = vmc_act_rt(oact, x);
  return vmc_dact_rt(oact, x, a) / a;
}

struct TailArgs {
  PackedParams pp;
  const float* z1;          // [n_base][Hp] cached first-layer pre-activations
  const float* on_base;     // [n_base] cached onsite term x . w_on (RBM) or nullptr
  const float* logit_base;  // [n_base] cached logits (ratio mode)
  const int2* rowinfo;      // [n_rows] {chain, signed bond+1 or 0}; never null (launch_iota_rows)
  const int2* bonds;        // [n_bonds] {i, j}
  const float* half_jx;     // [n_bonds] 0.5 * j_x
  const int* n_rows_dev;    // device row count (list mode) or nullptr
  int n_rows;               // host row count / upper bound
  int n_hidden;             // L-1
  int n_sites;              // N (rows of W1)
  int n_units;              // H (unpadded layer size; the RBM sum skips the padded units)
  int num_cus;              // CUs of the device (persistent grid size)
  int act;                  // hidden activation id (selects the instantiation)
  int oact;                 // output activation id (ratio mode: exp -> log domain, else linear)
  float* out;               // [n_rows]
};

struct SweepArgs {
  PackedParams pp;
  const float* configs_in;  // [B][N] +-1 chains at launch start
  const float* z1_in;       // [B][Hp] exact cache of configs_in (read when cache_in_valid)
  const float* logit_in;    // [B]
  float* configs;           // [B][N] chains at launch end (may alias configs_in)
  float* z1;                // [B][Hp] cache out
  float* logit;             // [B]     cache out
  float* onsite;            // [B]     cache out: x . w_on (RBM) or nullptr
  int rbm;                  // 1: RestrictedBoltzmannNetwork epilogue (log cosh + onsite term)
  unsigned long long* accepted;  // device counter (atomicAdd)
  const int* inj_up;        // injected proposals or nullptr
  const int* inj_dn;
  const float* inj_u;
  unsigned char* acc_mask;  // [B] out (last step) or nullptr
  int* dbg_up; int* dbg_dn; float* dbg_u;   // proposal dump (debug_proposals) or nullptr
  unsigned long long* dbg_cycles;           // [grid][4 waves][16 phases] -> diagnostic STAMP build
  int act, oact;            // hidden / output activation ids
  float* dact_out;          // [L][B][Hp] f'(z) of the final chains (cosine only) or nullptr
  int waves;                // waves per workgroup of the sweep kernel (4 or 8)
  int no_w1l;               // 1: never hold W1 in LDS (the two-workgroups-per-CU variant)
  int uh_lds;               // set by the launcher: the sampler's dedicated hand-over area is part of its LDS
  int cache_in_valid;       // z1 / logit already hold the exact cache of `configs`
  float* act_out;           // [L][B][Hp] activations of the final chains (gradient path) or nullptr
  int B, N, n_hidden;
  int chain_offset;
  uint32_t seed_lo, seed_hi;
  unsigned long long step0;
  long long n_steps;
  // bond census of the final chains (k_bond_count's job and arithmetic, eloc.hip) or cnt_out == nullptr:
  // the sampler has the chains in LDS when it ends, the launch that would count them costs 5 us
  const int2* bonds; const float* quarter_jz; int n_bonds;
  int* cnt_out; float* diag_out;
  const unsigned* p16s;     // split sampler only (CGS_VMC_SPLIT_BF16=2): the three-term weight image of tail_split.hip
};

// launchers (one per TU)
hipError_t launch_pack(hipStream_t s, const float* theta, int N, int H, int Hp,
                       const ParamLayout& lay, float* w1p, float* b1p, float* bh, float* p16,
                       float* p16t, float* woutp, float* bout, float* won);
// d logit / d z_l of every layer: act_all / delta_all are [n_hidden + 1][B][Hp]
// act: hidden activation id; dact_all: f'(z) arrays (cosine only) or nullptr; oscale: [B]
// (1/psi) d psi / d x of a non-exp output activation or nullptr
// eloc: the local-energy reduction of the same accumulate call rides in this launch (k_eloc_reduce's job,
// chain by chain with its summation order: the workgroup that owns 16 chains also folds their rows of
// `val`), one dependent launch (6 us at config 3) less per step; off == nullptr: nothing to fold
// ratio != nullptr (LogOverlapITSWO, training.py:665-672): eloc is the SUPERVISOR's local energy and the launch
// also leaves ratio_b = psi_w / psi (1 - beta E_loc^w) -- k_itswo_ratio's job and expression (grad.hip) -- from
// the two cached logits; that ratio is then the weight of the second sum (OutLayerSums::w == ratio)
struct ElocFold {
  const int* off; const float* diag; const float* val; float* offdiag; float* eloc;
  float* ratio; const float* logit_psi; const float* logit_omega; float log_factor, beta; int oact;
};
// out: the output layer's weight gradient (fully_connected: sum_b (1 | w_b) s_b [a_L(b) | 1], s = oscale or 1)
// as per-workgroup partial sums over the 16 chains a workgroup owns anyway, part[workgroup][2][Hp + 4]
// (slot Hp: the bias); k_wgrad folds them (its N = 1 tiles -- one column of 64 used -- leave the MFMA grid,
// which frees a k-slice).  w == ElocFold::eloc: the weights are the local energies this launch folds.
struct OutLayerSums { float* part; const float* w; };
hipError_t launch_backprop16(hipStream_t s, const float* act_all, float* delta_all,
                             const float* p16t, const float* woutp, int B, int Hp, int n_hidden,
                             bool rbm, int act, const float* dact_all, const float* oscale,
                             const ElocFold& eloc = ElocFold{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0},
                             const OutLayerSums& out = OutLayerSums{nullptr, nullptr});
hipError_t launch_tail(hipStream_t s, const TailArgs& a, int Hp, bool ratio_mode, bool rbm);
hipError_t launch_onsite(hipStream_t s, const float* configs, const float* won, int rows, int N,
                         float* out);
hipError_t launch_iota_rows(hipStream_t s, int2* dst, int n);
hipError_t launch_sweep16(hipStream_t s, const SweepArgs& a, int Hp);
// eight chains per workgroup (sweep8.hip): fully_connected + relu, plain launches, the shapes of plan_sweep8; the
// same chains as launch_sweep16, bit for bit
hipError_t launch_sweep8(hipStream_t s, const SweepArgs& a, int Hp);
// EXPERIMENT: the sampler with its H x H layers as 3 x bf16 split products (sweep_split.hip); false: shape not covered
bool sweep16_split_supported(int N, int Hp, int n_hidden);
hipError_t launch_sweep16_split(hipStream_t s, const SweepArgs& a);
// the LDS-operand row kernel for 384 / 512 padded units (tail_lds.hpp; one instantiation per hidden
// activation in act_tail.hip): fully_connected and rbm with at least one H x H layer
bool tail_lds_supported(int Hp, int n_hidden);
hipError_t launch_tail_lds(hipStream_t s, const TailArgs& a, int Hp, bool ratio_mode, bool rbm);
size_t sweep_lds_required(int N, int Hp, int n_hidden, bool rbm);
// the 3 x bf16 split experiment of the row kernel (tail_split.hip; CGS_VMC_SPLIT_BF16=1)
hipError_t launch_pack_split(hipStream_t s, const float* theta, int H, const ParamLayout& lay, unsigned* out);
long long pack_split_dwords(int n_hh);
hipError_t launch_tail16_split(hipStream_t s, const TailArgs& a, const unsigned* p16s, bool ratio_mode);

// bond list / local-energy reduction (eloc.hip)
hipError_t launch_bond_list(hipStream_t s, const float* configs, const int2* bonds,
                            const float* quarter_jz, int B, int N, int n_bonds, int* cnt,
                            int* off, float* diag, int2* rowinfo, bool counted = false);   // counted: cnt / diag are up to date
hipError_t launch_eloc_reduce(hipStream_t s, const int* off, const float* diag, const float* val,
                              int B, float* offdiag, float* eloc);
hipError_t launch_check_pm1(hipStream_t s, const float* x, long long n, int* flag);
hipError_t launch_sum(hipStream_t s, const float* x, int n, double* out_sum);
hipError_t launch_max(hipStream_t s, const float* x, int n, float* out_max);
hipError_t launch_div_f64(hipStream_t s, double* x, int n, double d);

// gradient path (grad.hip)
struct GemmArgs {
  const float* A; long long sam, sak;   // A(m,k) = A[m*sam + k*sak]
  const float* B; long long sbk, sbn;   // B(k,n) = B[k*sbk + n*sbn]
  const float* kscale;                  // optional: B(k,n) *= kscale[k] (dual: second product)
  int M, N, K;
  float* C; long long ldc;              // C(m,n) = C[m*ldc + n]
  float* C2;                            // dual: destination of A (kscale (.) B)
  int dual;                             // compute both A B -> C and A (kscale (.) B) -> C2
  int ones_row;                         // row M-1 of A is an implicit row of ones
  const float* bias;                    // epilogue 1: f(v + bias[n]), f = hidden activation `act`
  const float* mask;  long long ldmask; // epilogues 5 / 6: * f'(z) read off mask = f(z) (relu: mask > 0)
  int act;                              // hidden activation id of epilogues 1 / 5 / 6
  float* dact_out;                      // epilogue 1: also store f'(z) (layout of C) or nullptr
  int epilogue;                         // 0 none, 1 f(v + bias), 3 accumulate (C += ), 4 bias,
                                        // 5 f' (.) (v + bias), 6 f' (.) (C + v + bias), 7 tanh(v + bias), 8 C + v + bias,
                                        // 9 mask (.) (v + bias) with mask = a stored f'(z), 11 selu(v + bias),
                                        // 10 row dot: nothing is stored to C; dot_out[tn * M + m] = the sum over the 128
                                        //    columns of column tile tn of f(v + bias) * dot_w[n], in double (the output
                                        //    layer's dot product folded into the last H x H layer; only the 128 x 128-tile
                                        //    kernels have it: ask gemm_rowdot_ok)
  const float* dot_w; double* dot_out;  // epilogue 10
  // conv_a != 0 (k_gemm_ring only; gemm_conv_a_ok): A is NOT a matrix but a channel-last feature map [row][site][ca_Fp] of a
  // periodic convolution -- A(m = (row, site), k = (tap, c)) = map[row][(site + tap - lo) mod lattice][c]: the im2col
  // gather happens in the address of the kernel's A pieces (conv_general.hip).  K = taps x ca_F, ca_F % 32 == 0.
  int conv_a, ca_N, ca_D1, ca_D2, ca_KW, ca_lo, ca_lo2, ca_F, ca_Fp;
  int splitk;                           // >= 1
  float* workspace;                     // [splitk][dual ? 2 : 1][M][N] when splitk > 1
};
hipError_t launch_gemm(hipStream_t s, const GemmArgs& g);
// whether launch_gemm takes this product (epilogue 10 set) with a kernel that has the row-dot epilogue
bool gemm_rowdot_ok(const GemmArgs& g);
// whether launch_gemm takes this product with conv_a set (the implicit-gather form of k_gemm_ring)
bool gemm_conv_a_ok(const GemmArgs& g);
inline int gemm_rowdot_tiles(int N) { return (N + 127) / 128; }
// All weight gradients of one accumulate call -- [a_{l-1} | 1]^T [delta_l | w (.) delta_l] of every
// layer and the scalar accumulators -- in ONE launch (k_wgrad, grad.hip; tiles,
// slices and block order: plan.hpp).  The problem table lives in device memory and is built once per
// weight vector with wgrad_fill_problem into a host buffer of n x wgrad_problem_bytes().
struct WgradLaunch {
  const void* dev_problems; int n_prob;
  int tiles, slices;                   // plan_wgrad_total_tiles, plan_wgrad_slices
  int K;                               // samples
  const float* w;                      // [K] weights of the second sum
  float* g1; float* g2;                // accumulators (theta layout)
  float* ws; int* tickets;             // plan_wgrad_ws_floats(tiles, slices) floats, `tiles` zeroed ints
  bool fresh;                          // the accumulators hold no sum yet: store
  const float* sc_eloc; const float* sc_ratio; float* sc_out; int sc_B, sc_mode;   // scalar accumulators (sc_out may be null)
  // output layer from k_backprop16's partials (OutLayerSums) or nullptr: [out_nwg][2][out_ld] -> g1 / g2 + out_off
  const float* out_part; int out_nwg, out_H, out_ld; long long out_off;
};
size_t wgrad_problem_bytes();
void wgrad_fill_problem(void* dst, int index, const float* A, long long lda, const float* D, long long ldd,
                        long long c_off, int k_in, int n_out, int tile0);
hipError_t launch_wgrad(hipStream_t s, const WgradLaunch& L);
hipError_t launch_act_copy(hipStream_t s, const float* z, float* a, float* dact, long long n, int act);
hipError_t launch_out_scale(hipStream_t s, const float* x, float* oscale, int B, int oact);
hipError_t launch_tanh_copy(hipStream_t s, const float* z, float* a, long long n);
hipError_t launch_scalar_accum(hipStream_t s, const float* eloc, const float* ratio, int B,
                               float* acc_scalars, int mode, bool fresh = false);
hipError_t launch_itswo_ratio(hipStream_t s, const float* logit_psi, const float* logit_omega,
                              const float* eloc_omega, float log_factor, float beta, int B,
                              float* ratio, int oact);
hipError_t launch_adam(hipStream_t s, float* theta, float* m, float* v, const float* acc, int P,
                       int mode, float lr_t, float b1, float b2, float eps, float* grad_out);
hipError_t launch_fill(hipStream_t s, float* x, float v, long long n);
hipError_t launch_scale_one(hipStream_t s, float* x, float f);

// stochastic reconfiguration (sr.hip)
hipError_t launch_jvp_out(hipStream_t s, const float* tang, const float* act, const float* wout,
                          const float* vout, const float* vbout, int B, int H, int Hp, float* t);
hipError_t launch_jvp_out_rbm(hipStream_t s, const float* tang, const float* act,
                              const float* cfg, const float* von, const float* vbon, int B, int H,
                              int Hp, int N, float* t);
hipError_t launch_sum_into(hipStream_t s, const float* t, int B, float* dst);
hipError_t launch_sr_rhs(hipStream_t s, const float* acc, int P, float* x, float* r, float* p,
                         double* partial, double* sc);
hipError_t launch_sr_q(hipStream_t s, const float* u, const float* acc, int P, const float* p,
                       float lambda, float* q, double* partial, double* sc);
hipError_t launch_sr_step(hipStream_t s, double* sc, int cur, int P, float* p, const float* q,
                          float* x, float* r, double* partial);
hipError_t launch_sr_tsum(hipStream_t s, const float* t, int n, float* out);
hipError_t launch_sr_apply(hipStream_t s, float* theta, const float* x, float lr, int P);
// large-tile GEMMs of the SR matrix-vector product (srmm.hip)
hipError_t launch_sr_rowdot(hipStream_t s, const float* A, long long lda, const float* V,
                            long long ldv, const float* vb, const float* D, long long ldd, float* t,
                            int M, int N, int K, bool first);
int sr_wsum_slices(int R, int num_cus);
hipError_t launch_sr_wsum(hipStream_t s, const float* A, long long lda, const float* D,
                          long long ldd, const float* t, float* ws, float* out, long long ldo,
                          float* bias_out, int M, int N, int R, int slices);
// one row-dot problem: t(m) (= | +=) sum_n (A(m, :) V(:, n) + vb(n)) D(m, n)
struct SrRowdotArgs {
  const float* A; long long lda;      // A(m, k) = A[m * lda + k]: stored activations / chains
  const float* V; long long ldv;      // V(k, n) = V[k * ldv + n]: weight slice of the CG direction
  const float* vb;                    // [N] bias slice
  const float* D; long long ldd;      // delta(m, n)
  float* t;                           // [M]
  int M, N, K;
  int first;                          // 1: t = ..., 0: t += ...
};
hipError_t launch_sr_rowdot_batch(hipStream_t s, const SrRowdotArgs* probs, int count);
// t = ((parts_0 + parts_1) + ...) + (x . v + vb)   (nparts == 0: t += x . v + vb)
hipError_t launch_sr_row_linear(hipStream_t s, const float* x, long long ldx, const float* v,
                                const float* vb, int R, int K, float* t, const float* parts, int nparts,
                                long long pstride);
hipError_t launch_sr_colsum(hipStream_t s, const float* x, long long ldx, const float* t, int R,
                            int K, float* ws, int slices, float* u, float* tsum);

// fully_connected with fc_layer_size > 256: the general path through materialised rows (wide.hip)
hipError_t launch_wide_rows_act(hipStream_t s, const float* z1, const float* w1p, const int2* rowinfo,
                                const int2* bonds, long long row0, int n_rows, int Hp, int act, float* out);
// RBM onsite term of the rows of k_wide_out: x . w_on of the base configuration (`base`, per chain
// or per external row) plus the exchange update of the row's bond (`bonds`) or of the proposed
// move (`iup` / `idn`, one row per chain); base == nullptr: fully_connected
struct WideOnsite {
  const float* base; const float* won; const int2* bonds; const int* iup; const int* idn;
};
hipError_t launch_wide_out(hipStream_t s, const float* a, const float* wout, const float* bout, int n_rows,
                           int H, int Hp, const int2* rowinfo, long long row0, const float* half_jx,
                           const float* logit_base, int oact, bool ratio, float* out,
                           const WideOnsite& on = WideOnsite{nullptr, nullptr, nullptr, nullptr, nullptr});
// the same from the row-dot partials of the last H x H layer (GemmArgs epilogue 10): part[t * n_rows + r], t < n_part
hipError_t launch_wide_out_part(hipStream_t s, const double* part, int n_part, const float* bout, int n_rows,
                                const int2* rowinfo, long long row0, const float* half_jx, const float* logit_base,
                                int oact, bool ratio, float* out, const WideOnsite& on);
hipError_t launch_wide_propose(hipStream_t s, const float* configs, int B, int N, uint32_t seed_lo,
                               uint32_t seed_hi, int chain_offset, unsigned long long step, const int* inj_up,
                               const int* inj_dn, const float* inj_u, int* iup, int* idn, float* u);
// one mc_step of the general sampler in one launch (k_wide_step): accept the previous step's proposal from the
// last layer's activations `a_last`, propose this step's move, update z1, write the candidate's a0
struct WideStepArgs {
  float* configs; float* z1; const float* w1p;
  const float* a_last; float* a0;                 // may alias (see k_wide_step)
  const double* dot_part; int n_part;             // or the row-dot partials of the last layer [n_part][B] (a_last unused)
  const float* wout; const float* bout; float* logit;
  int* iup; int* idn; float* u;                   // the proposal in flight: read by (1), rewritten by (2)
  const int* inj_up; const int* inj_dn; const float* inj_u;   // injected proposal (tests) or nullptr
  unsigned long long* accepted; unsigned char* acc_mask;
  float* onsite; const float* won;                // RestrictedBoltzmannNetwork or nullptr
  int B, N, H, Hp, act, oact;
  int do_accept, do_propose;
  uint32_t seed_lo, seed_hi; int chain_offset; unsigned long long step;
};
hipError_t launch_wide_step(hipStream_t s, const WideStepArgs& a);
hipError_t launch_wide_delta_last(hipStream_t s, const float* a_last, const float* wout, const float* oscale,
                                  int B, int H, int Hp, int act, float* delta, const float* dact = nullptr);
