// Gradient-accumulator path: the two tf.gradients calls of training.py:545-547 / 674-679
// (sum_b O_k(b) and sum_b w_b O_k(b), O_k = d logit / d theta_k) as an explicit
// forward / back-prop (k_backprop16, mlp.hip) / weight-gradient GEMM chain on fp32 MFMA, plus the accumulator,
// ratio and Adam element-wise kernels.  All reductions are fixed-order (no float atomics).
// ---- arch guard: k_wgrad's split-K hand-over (sc1 write-through stores drained by every storing wave, a
// relaxed agent-scope ticket, sc1 loads by the last arriver) relies on the scope-bit behaviour measured on
// gfx950 (MI355X_MICROARCH.md, "Hand-offs measured with sc1 loads in place of the acquire"); it is not a
// C++-memory-model release/acquire pair.  Refuse to compile device code for anything else.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "grad.hip: k_wgrad's sc1 hand-over protocol is validated on gfx950 only"
#endif
#include "common.hpp"
#include <cstdlib>
#ifdef VMC_WGRAD_STAMPS
#include <algorithm>
#include <cstdio>
#include <vector>
#endif

// ----------------------------------------------------------------------------------- GEMM
// C[M,N] = A[M,K] B[K,N] with arbitrary element strides on fp32 MFMA.
//   64x64x32 block tile, 4 waves in a 2x2 grid, one (DUAL: two) 32x32 v_mfma_f32_32x32x2_f32
//   accumulator(s) per wave; global -> register prefetch of tile k+1 while tile k is computed
//   from LDS; dwordx4 global loads along whichever dimension has stride 1.
//   ones_row : C has one extra row M-1 = the column sums of B (the product with an implicit
//              row of ones): the bias gradient, which sits right behind its weight matrix in the
//              parameter vector.  It is accumulated on the VALU by the m-tile-0 workgroups from
//              the B tile they have staged anyway, so the MFMA grid only covers M-1 rows.
//   DUAL     : second product A (kscale (.) B) -> C2 from the same tiles (the weighted sum
//              sum_b w_b O_k next to sum_b O_k, training.py:545-547)
//   split-K over blockIdx.z into a workspace that k_gemm_reduce folds in z order.
#define GT 64
#define GK 32
#define GLD 68

__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, float* C, int m, int n, float v) {
  float* c = C + (long long)m * g.ldc + n;
  switch (g.epilogue) {
    case 1: {   // hidden layer: a = f(z) (and f'(z) next to it where the activation needs z: cosine)
      const float z = v + g.bias[n];
      const float a = vmc_act_rt(g.act, z);
      *c = a;
      if (g.dact_out) g.dact_out[(long long)m * g.ldc + n] = vmc_dact_rt(g.act, z, a);
      break;
    }
    case 3: *c += v; break;
    case 4: *c = v + g.bias[n]; break;
    // tangent pass: f'(z) (.) (...), f' read off the stored activation a = f(z)
    case 5: *c = vmc_dact_rt(g.act, g.mask[(long long)m * g.ldmask + n], g.mask[(long long)m * g.ldmask + n]) * (v + g.bias[n]); break;
    case 7: *c = tanhf(v + g.bias[n]); break;
    case 9: *c = g.mask[(long long)m * g.ldmask + n] * (v + g.bias[n]); break;   // mask IS f'(z) (cosine: stored by epilogue 1)
    case 8: *c = *c + v + g.bias[n]; break;
    case 11: {  // ResBlock2d's selu (layers.py:226) -- the general convolution path stores selu(first_conv(h))
      const float z = v + g.bias[n];
      *c = 1.0507009873554805f * (z > 0.f ? z : 1.6732632423543772f * (expf(z) - 1.f));
      break;
    }
    case 6: *c = vmc_dact_rt(g.act, g.mask[(long long)m * g.ldmask + n], g.mask[(long long)m * g.ldmask + n]) * (*c + v + g.bias[n]); break;
    default: *c = v; break;
  }
}

// One operand tile (64 along d, 32 along k) as 2 x float4 per thread.
//   dfast (stride_d == 1): i-th vector = elements (d0 + 4*(t%16) + j, k0 + t/16 + 16 i)
//   kfast (stride_k == 1): i-th vector = elements (d0 + t/8 + 32 i, k0 + 4*(t%8) + j)
struct TileRegs { f32x4 v[2]; };

__device__ __forceinline__ TileRegs load_tile(const float* __restrict__ P, long long sd,
                                              long long sk, int d_extent, int kend, int d0,
                                              int k0, bool kfast, int ones_d, int t) {
  TileRegs r;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int d, k, dd, dk;
    if (kfast) { d = d0 + (t >> 3) + 32 * i; k = k0 + 4 * (t & 7); dd = 0; dk = 1; }
    else       { d = d0 + 4 * (t & 15); k = k0 + (t >> 4) + 16 * i; dd = 1; dk = 0; }
    const float* p = P + (long long)d * sd + (long long)k * sk;
    const bool inb = (d + 3 * dd < d_extent) && (k + 3 * dk < kend);
    const long long sv = kfast ? sk : sd;
    if (inb && sv == 1 && (((size_t)p) & 15) == 0) {
      r.v[i] = *(const f32x4*)p;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int dj = d + j * dd, kj = k + j * dk;
        r.v[i][j] = (dj < d_extent && kj < kend) ? P[(long long)dj * sd + (long long)kj * sk] : 0.f;
      }
    }
    if (ones_d >= 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (d + j * dd == ones_d && k + j * dk < kend) r.v[i][j] = 1.f;
    }
  }
  return r;
}

__device__ __forceinline__ void store_tile(float (*S)[GLD], const TileRegs& r, bool kfast, int t) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (kfast) {
      const int d = (t >> 3) + 32 * i, k = 4 * (t & 7);
#pragma unroll
      for (int j = 0; j < 4; ++j) S[k + j][d] = r.v[i][j];
    } else {
      const int d = 4 * (t & 15), k = (t >> 4) + 16 * i;
      *(f32x4*)&S[k][d] = r.v[i];
    }
  }
}

__device__ __forceinline__ TileRegs scale_tile(const TileRegs& r, const float* __restrict__ ks,
                                               int k0, int kend, bool kfast, int t) {
  TileRegs o;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = kfast ? k0 + 4 * (t & 7) + j : k0 + (t >> 4) + 16 * i;
      o.v[i][j] = k < kend ? r.v[i][j] * ks[k] : 0.f;
    }
  }
  return o;
}

// kscale values of one B tile.  dfast tile: element (i, j) has k = k0 + t/16 + 16 i -> s[i];
// kfast tile: k = k0 + 4 (t%8) + j -> s[j]
struct ScaleRegs { f32x4 s; };

__device__ __forceinline__ ScaleRegs load_scale(const float* __restrict__ ks, int k0, int kend,
                                                bool kfast, int t) {
  ScaleRegs o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = kfast ? k0 + 4 * (t & 7) + j : k0 + (t >> 4) + 16 * (j & 1);
    o.s[j] = ks[k < kend ? k : kend - 1];
  }
  return o;
}

__device__ __forceinline__ TileRegs apply_scale(const TileRegs& r, const ScaleRegs& sc, bool kfast) {
  TileRegs o;   // out-of-range k were loaded as 0 already
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (kfast) o.v[i] = r.v[i] * sc.s;
    else o.v[i] = r.v[i] * sc.s[i];
  }
  return o;
}

// (A second register stage -- tile t+2 in flight while tile t is multiplied -- was measured and
// gives nothing: 19.7 vs 18.2 us on the 256-workgroup back-prop GEMMs; NS stays 1.)
template <bool DUAL, int NS = 1>
__device__ __forceinline__ void gemm_block(const GemmArgs& g, int bx, int by, int bz) {
  __shared__ __attribute__((aligned(16))) float As[GK][GLD];
  __shared__ __attribute__((aligned(16))) float Bs[GK][GLD];
  __shared__ __attribute__((aligned(16))) float Bs2[DUAL ? GK : 1][GLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = by * GT, n0 = bx * GT;
  int kc = (g.K + g.splitk - 1) / g.splitk;
  kc = (kc + GK - 1) / GK * GK;
  const int kbeg = bz * kc;
  const int kend = min(g.K, kbeg + kc);
  const bool a_kfast = (g.sak == 1), b_kfast = (g.sbk == 1);
  const int a_extent = g.ones_row ? g.M - 1 : g.M;
  const bool do_colsum = g.ones_row && by == 0 && tid < GT;
  const bool scaled = g.kscale != nullptr;
  double cs = 0.0, cs2 = 0.0;      // (the ones row: a plain column sum over K, carried in double)

  f32x16 acc, acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }

  // Tile t+NS is requested right after tile t has gone to LDS, i.e. NS compute phases (16 MFMAs
  // per wave each, twice that for DUAL) before it is needed.  The requests are unconditional
  // (tile index clamped) so that vmcnt is counted exactly.
  const int T = kbeg < kend ? (kend - kbeg + GK - 1) / GK : 0;
  TileRegs ra[NS], rb[NS];
  ScaleRegs rs[NS];
  auto request = [&](int t, TileRegs& a, TileRegs& b, ScaleRegs& sc) {
    const int k0 = kbeg + min(t, T - 1) * GK;
    a = load_tile(g.A, g.sam, g.sak, a_extent, kend, m0, k0, a_kfast, -1, tid);
    b = load_tile(g.B, g.sbn, g.sbk, g.N, kend, n0, k0, b_kfast, -1, tid);
    if (scaled) sc = load_scale(g.kscale, k0, kend, b_kfast, tid);
  };
  auto stage = [&](int t, TileRegs& a, TileRegs& b, ScaleRegs& sc) {
    store_tile(As, a, a_kfast, tid);
    if (DUAL) {
      store_tile(Bs, b, b_kfast, tid);
      store_tile(Bs2, apply_scale(b, sc, b_kfast), b_kfast, tid);
    } else if (scaled) {
      store_tile(Bs, apply_scale(b, sc, b_kfast), b_kfast, tid);
    } else {
      store_tile(Bs, b, b_kfast, tid);
    }
    __syncthreads();
    request(t + NS, a, b, sc);
    if (do_colsum) {
#pragma unroll
      for (int kk = 0; kk < GK; ++kk) {
        cs += (double)Bs[kk][tid];
        if (DUAL) cs2 += (double)Bs2[kk][tid];
      }
    }
#pragma unroll
    for (int kk = 0; kk < GK; kk += 2) {
      const float av = As[kk + (lane >> 5)][wm * 32 + (lane & 31)];
      const float bv = Bs[kk + (lane >> 5)][wn * 32 + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
      if (DUAL) {
        const float bv2 = Bs2[kk + (lane >> 5)][wn * 32 + (lane & 31)];
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv2, acc2, 0, 0, 0);
      }
    }
    __syncthreads();
  };
  if (T > 0) {
#pragma unroll
    for (int i = 0; i < NS; ++i) request(i, ra[i], rb[i], rs[i]);
    for (int t = 0; t < T; t += NS) {
#pragma unroll
      for (int i = 0; i < NS; ++i)
        if (i == 0 || t + i < T) stage(t + i, ra[i], rb[i], rs[i]);
    }
  }

  const int n = n0 + wn * 32 + (lane & 31);
  const long long mn = (long long)g.M * g.N;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (m < a_extent && n < g.N) {
      if (g.splitk > 1) {
        float* ws = g.workspace + (long long)bz * (DUAL ? 2 : 1) * mn;
        ws[(long long)m * g.N + n] = acc[r];
        if (DUAL) ws[mn + (long long)m * g.N + n] = acc2[r];
      } else {
        gemm_epilogue(g, g.C, m, n, acc[r]);
        if (DUAL) gemm_epilogue(g, g.C2, m, n, acc2[r]);
      }
    }
  }
  if (do_colsum && n0 + tid < g.N) {
    const int m = g.M - 1, nn = n0 + tid;
    if (g.splitk > 1) {
      float* ws = g.workspace + (long long)bz * (DUAL ? 2 : 1) * mn;
      ws[(long long)m * g.N + nn] = (float)cs;
      if (DUAL) ws[mn + (long long)m * g.N + nn] = (float)cs2;
    } else {
      gemm_epilogue(g, g.C, m, nn, (float)cs);
      if (DUAL) gemm_epilogue(g, g.C2, m, nn, (float)cs2);
    }
  }
}

template <bool DUAL, int NS = 1>
__global__ __launch_bounds__(256) void k_gemm(GemmArgs g) {
  gemm_block<DUAL, NS>(g, blockIdx.x, blockIdx.y, blockIdx.z);
}

// tf.metrics.mean updates (training.py:555, 689-690) + mean_tensor count (550-553):
// scalars = [e_total, e_count, r_total, r_count, g_count]; one workgroup, fixed order
struct ScalarJob { const float* eloc; const float* ratio; float* sc; int B, mode; };

__device__ __forceinline__ void scalar_accum_body(const ScalarJob& j, bool fresh, double* se, double* sr) {
  double e = 0.0, r = 0.0;
  for (int i = threadIdx.x; i < j.B; i += 256) {
    e += (double)j.eloc[i];
    if (j.ratio) r += (double)j.ratio[i];
  }
  se[threadIdx.x] = e; sr[threadIdx.x] = r;
  __syncthreads();
  for (int d = 128; d >= 1; d >>= 1) {
    if ((int)threadIdx.x < d) { se[threadIdx.x] += se[threadIdx.x + d]; sr[threadIdx.x] += sr[threadIdx.x + d]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float* sc = j.sc;
    if (fresh) {   // first accumulate after reset_gradients: the scalars hold no sum yet
#pragma unroll
      for (int i = 0; i < 8; ++i) sc[i] = 0.f;
    }
    sc[0] += (float)se[0];
    sc[1] += (float)j.B;
    if (j.mode == 1) { sc[2] += (float)sr[0]; sc[3] += (float)j.B; }
    sc[4] += 1.f;
  }
}

// ----------------------------------------------------------------- batched weight-gradient kernel
// Every weight gradient of the dense ansatz types is [a_{l-1} | 1]^T [delta_l | w (.) delta_l] over the
// B samples of the batch (training.py:545-547): rows 0 .. k_in-1 give dW, the implicit ones row gives
// db (b_l sits right behind w_l in theta); the unscaled product goes to g1, the w-scaled one to g2.
// ALL layers run as ONE launch of two kinds of workgroup (block-uniform dispatch on the block id):
//  * MFMA tiles: a 64 x 64 output tile of one layer over one K slice of the samples.  4 waves as 2 x 2,
//    each one 32x32x2 accumulator per product; the operands of k-tile t+1 travel global -> registers
//    while tile t is multiplied from LDS and go to the OTHER LDS buffer afterwards: one barrier per 32
//    samples.  (128 x 64 tiles halve the operand traffic per flop but double the partial a last arriver
//    has to read from other XCDs -- 62-70 GB/s per workgroup, MI355X_MICROARCH.md -- which is the
//    critical path of the tail: 64 x 64 with half as many slices.)  Slices are sized so that tiles x slices
//    fills the CUs once (plan_wgrad_slices); slice s lives on XCD s % 8, whose L2 then fetches that
//    slice's operand rows once (plan_wgrad_block).  The partial tile goes to the workspace in
//    accumulator order (coalesced), the workgroup takes a ticket, and the LAST one to arrive folds the
//    slices 0 .. S-1 of its tile in that fixed order into the accumulators -- no reduction launch, no
//    float atomics, the same bits whoever arrives last.
//    The N = 1 problems (output layer: d logit / d w_out = a_L; RBM onsite layer) are one-column tiles.
//  * one workgroup for the scalar accumulators (sum E, counts) of the same accumulate call.
struct WgradProblem {
  const float* A; long long lda;      // A(b, m) = A[b * lda + m], m < k_in (m == k_in: the ones row)
  const float* D; long long ldd;      // delta(b, n) = D[b * ldd + n], n < n_out
  long long c_off;                    // offset of C[(k_in + 1)][n_out] in g1 / g2 (theta layout: w then b)
  int k_in, n_out;
  int tile0, tiles_n;                 // first MFMA tile of this problem, tiles along n
};

struct WgradArgs {
  const WgradProblem* prob;           // device table
  int n_prob;
  int tiles, slices, kchunk;          // MFMA tiles over all problems, K slices, samples per slice
  int mfma_blocks;                    // block ids [0, mfma_blocks): tiles; the next one: the scalars
  int K;                              // samples
  const float* w;                     // [K] per-sample weight of the second sum
  float* g1; float* g2;
  float* ws;                          // [tiles][slices][2 x WG_TM x WG_TN + 2 x WG_TN] partial tiles + ones rows
  int* tickets;                       // [tiles], zero between launches
  int fresh;                          // 1: the accumulators hold no sum yet (store instead of add)
  unsigned long long* stamps;         // diagnostic build (-DVMC_WGRAD_STAMPS): [blocks][8] wall-clock stamps (100 MHz)
  ScalarJob job;
  const float* out_part; int out_nwg, out_H, out_ld; long long out_off;   // WgradLaunch: the output layer's partials
};

// LDS image of one k-tile: A and delta TRANSPOSED, [m or n][position of k], row stride WG_LDT floats; the
// 32 samples of a tile sit at position p(k) = 16 (k & 1) + (k >> 1), so that the 16 operands an MFMA lane
// needs over the tile (32x32x2: lane (row, half h) takes k = 2 q + h) are 16 CONSECUTIVE floats = four
// ds_read_b128 instead of sixteen ds_read_b32 (a single wave per SIMD gets a fifth of the ds_read_b32
// rate: MI355X_MICROARCH.md, LDS).  WG_LDT = 36: rows stay 16-byte aligned and the 16 lanes of a
// ds_read_b128 group start on 16 distinct bank quads (36 / 4 = 9 is odd).
#define WG_LDT 36
#define WG_STAGE ((WG_TM + WG_TN) * WG_LDT + WG_TK)          // floats of one LDS stage: A^T, delta^T, w
#define WG_QUADS 8                                           // accumulator quads per thread (2 products x 16 / 4)
#define WG_PART (WG_TM * WG_TN * 2 + 2 * WG_TN)              // floats of one partial: both products' tiles + ones rows

// Hand-over of the partial tiles between workgroups (other CUs, other XCDs: their L2s are not coherent).
// MI355X_MICROARCH.md "inter-workgroup visibility", the measured row "ONE lane of each storing workgroup
// adds to ONE counter; the workgroup whose add came last loads": EVERY store of the handed-over bytes is a
// 16-byte sc1 (write-through) store, every storing wave drains them (s_waitcnt vmcnt(0)) before the
// workgroup barrier behind which lane 0 takes the ticket with an agent-scope atomic add; the last arriver
// loads EVERY byte with 16-byte sc1 loads behind a barrier its ticket-taking wave has joined.  One
// workgroup per CU (the launch asks for more than half a CU's LDS).  A release fence instead
// (buffer_wbl2) costs microseconds per wave: the first version of this kernel took 250 us that way.
typedef __attribute__((address_space(1))) f32x4* wg_gf4_p;
__device__ __forceinline__ void wg_store_sc1(float* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"((wg_gf4_p)p), "v"(v) : "memory");
}
// quads of up to 8 slices z0 .. z0 + 7 (clamped to the last slice: unconditional loads), all in flight at once
__device__ __forceinline__ void wg_load8_sc1(const float* p0, long long stride, int z0, int S, f32x4 (&v)[8]) {
  wg_gf4_p q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) q[j] = (wg_gf4_p)(p0 + (long long)min(z0 + j, S - 1) * stride);
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc1\n\t"
      "global_load_dwordx4 %1, %9, off sc1\n\t"
      "global_load_dwordx4 %2, %10, off sc1\n\t"
      "global_load_dwordx4 %3, %11, off sc1\n\t"
      "global_load_dwordx4 %4, %12, off sc1\n\t"
      "global_load_dwordx4 %5, %13, off sc1\n\t"
      "global_load_dwordx4 %6, %14, off sc1\n\t"
      "global_load_dwordx4 %7, %15, off sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7])
      : "memory");
}
// sum over the slices 0 .. S-1, in that order, of one quad
__device__ __forceinline__ f32x4 wg_fold_quad(const float* p0, long long stride, int S) {
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  for (int z0 = 0; z0 < S; z0 += 8) {
    f32x4 v[8];
    wg_load8_sc1(p0, stride, z0, S, v);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (z0 + j < S) sum += v[j];
  }
  return sum;
}

// the same for two quads at once: sixteen loads in flight
__device__ __forceinline__ void wg_fold_quads2(const float* pa, const float* pb, long long stride, int S, f32x4& sa, f32x4& sb) {
  sa = f32x4{0.f, 0.f, 0.f, 0.f};
  sb = sa;
  for (int z0 = 0; z0 < S; z0 += 8) {
    wg_gf4_p qa[8], qb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long long o = (long long)min(z0 + j, S - 1) * stride;
      qa[j] = (wg_gf4_p)(pa + o);
      qb[j] = (wg_gf4_p)(pb + o);
    }
    f32x4 va[8], vb[8];
    asm volatile(
        "global_load_dwordx4 %0, %16, off sc1\n\t"
        "global_load_dwordx4 %1, %17, off sc1\n\t"
        "global_load_dwordx4 %2, %18, off sc1\n\t"
        "global_load_dwordx4 %3, %19, off sc1\n\t"
        "global_load_dwordx4 %4, %20, off sc1\n\t"
        "global_load_dwordx4 %5, %21, off sc1\n\t"
        "global_load_dwordx4 %6, %22, off sc1\n\t"
        "global_load_dwordx4 %7, %23, off sc1\n\t"
        "global_load_dwordx4 %8, %24, off sc1\n\t"
        "global_load_dwordx4 %9, %25, off sc1\n\t"
        "global_load_dwordx4 %10, %26, off sc1\n\t"
        "global_load_dwordx4 %11, %27, off sc1\n\t"
        "global_load_dwordx4 %12, %28, off sc1\n\t"
        "global_load_dwordx4 %13, %29, off sc1\n\t"
        "global_load_dwordx4 %14, %30, off sc1\n\t"
        "global_load_dwordx4 %15, %31, off sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(va[0]), "=&v"(va[1]), "=&v"(va[2]), "=&v"(va[3]), "=&v"(va[4]), "=&v"(va[5]), "=&v"(va[6]), "=&v"(va[7]),
          "=&v"(vb[0]), "=&v"(vb[1]), "=&v"(vb[2]), "=&v"(vb[3]), "=&v"(vb[4]), "=&v"(vb[5]), "=&v"(vb[6]), "=&v"(vb[7])
        : "v"(qa[0]), "v"(qa[1]), "v"(qa[2]), "v"(qa[3]), "v"(qa[4]), "v"(qa[5]), "v"(qa[6]), "v"(qa[7]),
          "v"(qb[0]), "v"(qb[1]), "v"(qb[2]), "v"(qb[3]), "v"(qb[4]), "v"(qb[5]), "v"(qb[6]), "v"(qb[7])
        : "memory");
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (z0 + j < S) { sa += va[j]; sb += vb[j]; }
  }
}

struct WgradRegs { float a[8]; float b[8]; float w; };

typedef const __attribute__((address_space(1))) float* wg_gfc_p;

// Loading the operands of a k-tile (the lambdas of wgrad_tile) is BRANCH-FREE and in the global address
// space: a load under a run-time branch makes the compiler drain the whole memory queue at the join (the
// first version -- `if (aligned) vector load else scalar loads` on flat pointers -- serialised five round
// trips per k-tile behind `s_waitcnt vmcnt(0)`: 137 us per launch).  Out-of-range samples, rows and
// columns are read from a clamped, valid address and zeroed by a select WHEN THEY GO TO LDS: a select at
// load time lets the compiler wait for every load of the ring at once.  Lane l of wave w takes row m0 + l
// of A (column n0 + l of delta) at the samples 4 c + e of the k-quads c = w, w + 4: 64 consecutive floats
// per load instruction, any stride, any alignment.
// accumulator register r (0 .. 15) of a thread -> (m, n) inside the 64 x 64 tile (32x32x2 MFMA layout)
__device__ __forceinline__ void wgrad_mn(int tid, int r, int& m, int& n) {
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
  n = wn * 32 + (lane & 31);
}

#ifdef VMC_WGRAD_STAMPS
#define WG_STAMP(i) do { if (a.stamps && threadIdx.x == 0) a.stamps[(long long)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define WG_STAMP(i) do { } while (0)
#endif

// FAST: every k-tile of the slice is full (the slice holds a whole number of 32-sample tiles; all slices
// but possibly the last do): no per-element clamping, ONE scalar base per matrix and tile.
template <bool FAST>
__device__ __forceinline__ void wgrad_tile(const WgradArgs& a, const WgradProblem& P, int slice, int tile, float* smem) {
  // Two GROUPS of four waves share the tile: group g takes the first / second half of the slice's k-tiles
  // with its own LDS stages and accumulators, group 1's sums are added to group 0's at the end (always in
  // that order).  Two waves per SIMD with the same instruction stream fill each other's gaps -- LDS read
  // latency after every barrier, the VALU / LDS / VMEM pieces between the MFMAs: with one wave per SIMD a
  // k-tile took 3,076 cycles for 2,048 cycles of MFMAs (s_memtime stamps, tools/wgrad_stamps.sh).
  const int grp = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  smem += grp * 3 * WG_STAGE;
  WG_STAMP(0);
  const int lt = tile - P.tile0, tm = lt / P.tiles_n, tn = lt % P.tiles_n;
  const int m0 = tm * WG_TM, n0 = tn * WG_TN;
  const int s_kbeg = slice * a.kchunk, s_kend = min(a.K, s_kbeg + a.kchunk);
  const int s_T = (s_kend - s_kbeg + WG_TK - 1) / WG_TK;       // >= 1: no slice is empty (plan_wgrad_slices)
  const int T0 = (s_T + 1) / 2;                                // k-tiles of group 0; group 1: the rest (maybe none)
  const bool idle = grp == 1 && s_T - T0 == 0;                 // nothing to do: reads the slice's first tile, stages zeros
  const int kbeg = (grp == 0 || idle) ? s_kbeg : s_kbeg + T0 * WG_TK;
  const int kend = grp == 0 ? min(s_kend, s_kbeg + T0 * WG_TK) : s_kend;
  const int T = (kend - kbeg + WG_TK - 1) / WG_TK;             // >= 1
  const int T_loop = T0;                                       // both groups make the same number of barriers
  const bool ones = tm == 0;                                   // the m-tile-0 workgroups own the ones row (column sums of delta)
  const bool sums = ones && __builtin_amdgcn_readfirstlane(wm) == 0;   // ... and their waves 0, 1 hold every column of delta once

  f32x16 acc, acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
  float cs = 0.f, cs2 = 0.f;

  // Three k-tiles of operands are in flight global -> registers (a ring of three register sets) while the
  // tile before them sits in LDS and the one before that is multiplied: with ONE workgroup per CU (one
  // wave per SIMD) nothing else hides the 1 - 2 us a load takes under load; one tile of look-ahead
  // (0.85 us of MFMAs) left every iteration waiting for memory (56 us per launch against 22 of MFMAs).
  // The loop is unrolled three times so that ring slot and LDS stage are compile-time: straight-line
  // code, every load unconditional (tiles past the end of the slice are clamped reads that stage
  // zeros), so vmcnt is counted exactly and nothing drains the queue.
  const int h16 = (lane >> 5) * 16;
  WgradRegs R0, R1, R2;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const bool m_in = m0 + lane < P.k_in, n_in = n0 + lane < P.n_out;
  // Addresses: a SCALAR row base (the sample index of a load is wave-uniform: SALU multiplies) plus a
  // 32-bit lane offset that never changes -- `global_load_dword v, v_off, s[base]`.  Per-lane 64-bit
  // products (two quarter-rate v_mul_lo_u32 + v_mad_u64_u32 per load, seventeen loads per k-tile) cost
  // 1,600 VALU cycles per 2,048-cycle tile, all of them paid in matrix time.
  const int swave = __builtin_amdgcn_readfirstlane(wave);
  const int a_off = m_in ? m0 + lane : 0, d_off = n_in ? n0 + lane : 0;
  wg_gfc_p pw = (wg_gfc_p)a.w;
  typedef const __attribute__((address_space(1))) char* wg_gcc_p;
  // FAST: byte offsets of the eight elements of a register set from the tile's scalar base -- loop
  // invariant, 32 bits: one `global_load_dword v, v_off, s[base:base+1]` per element and nothing else.
  // (The general form below computes a clamped row per element: 14 SALU instructions per load, 240 per
  // k-tile; with one wave per SIMD the kernel was bound by instruction ISSUE -- ~430 instructions per
  // 2,048-cycle tile at 8 cycles each -- not by the matrix pipe: 3,430 cycles per tile measured.)
  unsigned avo[8], dvo[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = 16 * (j >> 2) + (j & 3);
    avo[j] = 4u * (unsigned)(a_off + row * (int)P.lda);
    dvo[j] = 4u * (unsigned)(d_off + row * (int)P.ldd);
  }
  const unsigned wvo = 4u * (unsigned)(tid & 31);
  // scalar bases of tile tt (FAST): tiles past the end re-read the last one (their values are zeroed)
  wg_gcc_p sA = nullptr, sD = nullptr, sW = nullptr;
  auto bases = [&](int tt) {
    const int tc = min(tt, T - 1);
    const long long k0 = kbeg + tc * WG_TK;
    sA = (wg_gcc_p)P.A + 4 * (k0 + 4 * swave) * P.lda;
    sD = (wg_gcc_p)P.D + 4 * (k0 + 4 * swave) * P.ldd;
    sW = (wg_gcc_p)a.w + 4 * k0;
  };
  // element j (0 .. 7) of a register set: sample 4 (wave + 4 (j >> 2)) + (j & 3) of its tile
  auto load1 = [&](WgradRegs& r, int tt, int j) {
    if (FAST) {
      r.a[j] = *(wg_gfc_p)(sA + avo[j]);
      r.b[j] = *(wg_gfc_p)(sD + dvo[j]);
      return;
    }
    const int k = kbeg + tt * WG_TK + 4 * (swave + 4 * (j >> 2)) + (j & 3);
    const int kc = k < kend ? k : kend - 1;
    wg_gfc_p ra = (wg_gfc_p)P.A + (long long)kc * P.lda;
    wg_gfc_p rd = (wg_gfc_p)P.D + (long long)kc * P.ldd;
    r.a[j] = ra[a_off];
    r.b[j] = rd[d_off];
  };
  auto loadw = [&](WgradRegs& r, int tt) {
    if (FAST) { r.w = *(wg_gfc_p)(sW + wvo); return; }
    const int k0 = kbeg + tt * WG_TK;                       // scalar; at least one sample of the slice lies at or before it
    const int kk = min(tid & 31, kend - 1 - min(k0, kend - 1));
    r.w = (pw + min(k0, kend - 1))[kk];
  };
  // store j (0 .. 7) of a register set: matrix (j >> 2), k-quad wave + 4 ((j >> 1) & 1), parity (j & 1):
  // samples 4c + parity and 4c + parity + 2 go to positions 16 parity + 2c, + 1
  auto store1 = [&](float* st, const WgradRegs& r, int tt, int j) {
    const int x = j >> 2, i = (j >> 1) & 1, par = j & 1, c = swave + 4 * i;
    const int k = kbeg + tt * WG_TK + 4 * c + par;
    const bool in = x ? n_in : m_in;
    const float* v = x ? r.b : r.a;
    const bool l0 = !idle && (FAST ? tt < T : k < kend), l1 = !idle && (FAST ? tt < T : k + 2 < kend);
    const float v0 = (l0 && in) ? v[4 * i + par] : 0.f;
    const float v1 = (l1 && in) ? v[4 * i + par + 2] : 0.f;
    *(f32x2*)(st + x * WG_TM * WG_LDT + lane * WG_LDT + 16 * par + 2 * c) = f32x2{v0, v1};
  };
  // every thread writes w (eight of them the same value to each of the 32 positions): under `if (tid < 32)`
  // the compiler sinks the load of w into the branch and drains the whole memory queue behind it
  auto storew = [&](float* st, const WgradRegs& r, int tt) {
    const bool live = !idle && (FAST ? tt < T : kbeg + tt * WG_TK + (tid & 31) < kend);
    st[(WG_TM + WG_TN) * WG_LDT + 16 * (tid & 1) + ((tid & 31) >> 1)] = live ? r.w : 0.f;
  };
  // One k-tile: 32 MFMAs from stage `cur`; BETWEEN them, one piece per MFMA pair, the register set `r`
  // (tile tt + 1) goes to stage `nxt` (pieces 0 .. 8) and is then refilled with tile tt + 4 (pieces
  // 8 .. 15).  With one wave per SIMD an instruction only overlaps an MFMA if it is issued right behind
  // it: compute, THEN stage, THEN load cost 13.7 + 7 + 26 us per launch, each in full
  // (CGS_VMC_WGRAD_DBG experiments).  sched_barrier pins the interleaving.
  auto step = [&](const float* cur, float* nxt, WgradRegs& r, int tt) {
    const f32x4* At = (const f32x4*)(cur + (wm * 32 + (lane & 31)) * WG_LDT + h16);
    const f32x4* Dt = (const f32x4*)(cur + WG_TM * WG_LDT + (wn * 32 + (lane & 31)) * WG_LDT + h16);
    const f32x4* Wt = (const f32x4*)(cur + (WG_TM + WG_TN) * WG_LDT + h16);
    f32x4 av[4], bv[4], wv[4], b2v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { av[q] = At[q]; bv[q] = Dt[q]; wv[q] = Wt[q]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) b2v[q] = bv[q] * wv[q];
    if (FAST) bases(tt + 4);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q >> 2][q & 3], bv[q >> 2][q & 3], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q >> 2][q & 3], b2v[q >> 2][q & 3], acc2, 0, 0, 0);
      if (q < 8) store1(nxt, r, tt + 1, q);
      if (q == 8) storew(nxt, r, tt + 1);
      if (q >= 8) load1(r, tt + 4, q - 8);
      if (q == 15) loadw(r, tt + 4);
      if (sums) {     // block- and wave-uniform; fixed order; one plain v_add each (left to itself the compiler
                      // SLP-packs them behind v_mov pairs: three VALU instructions per add, beside MFMAs)
        asm volatile("v_add_f32 %0, %1, %0" : "+v"(cs) : "v"(bv[q >> 2][q & 3]));
        asm volatile("v_add_f32 %0, %1, %0" : "+v"(cs2) : "v"(b2v[q >> 2][q & 3]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // Three k-tiles of operands are in flight global -> registers (a ring of three register sets) while the
  // tile before them sits in LDS and the one before that is multiplied.  The loop is unrolled three times
  // so that ring slot and LDS stage are compile-time: straight-line code, every load unconditional (tiles
  // past the end of the slice are clamped reads that stage zeros), so vmcnt is counted exactly.
  float* L0 = smem;
  float* L1 = smem + WG_STAGE;
  float* L2 = smem + 2 * WG_STAGE;
  auto load_all = [&](WgradRegs& r, int tt) {
    if (FAST) bases(tt);
#pragma unroll
    for (int j = 0; j < 8; ++j) load1(r, tt, j);
    loadw(r, tt);
  };
  load_all(R0, 0); load_all(R1, 1); load_all(R2, 2);
#pragma unroll
  for (int j = 0; j < 8; ++j) store1(L0, R0, 0, j);
  storew(L0, R0, 0);
  load_all(R0, 3);
  __syncthreads();
  WG_STAMP(1);
#ifdef VMC_WGRAD_STAMPS
  if (a.stamps && threadIdx.x == 0) a.stamps[(long long)blockIdx.x * 8 + 6] = clock64();
#endif
  for (int t = 0; t < T_loop; t += 3) {     // tiles t, t + 1, t + 2 (those >= T are zeros)
    step(L0, L1, R1, t); __syncthreads();
    step(L1, L2, R2, t + 1); __syncthreads();
    step(L2, L0, R0, t + 2); __syncthreads();
  }
  WG_STAMP(2);
#ifdef VMC_WGRAD_STAMPS
  if (a.stamps && threadIdx.x == 0) a.stamps[(long long)blockIdx.x * 8 + 7] = clock64();
#endif
  // group 1 hands its sums to group 0 through its own (now idle) LDS stages
  {
    float* g1s = grp ? smem : smem + 3 * WG_STAGE;     // group 1's stages (smem already points there for group 1)
    if (grp) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { g1s[r * 256 + tid] = acc[r]; g1s[(16 + r) * 256 + tid] = acc2[r]; }
      g1s[32 * 256 + tid] = cs;
      g1s[33 * 256 + tid] = cs2;
    }
    __syncthreads();
    if (!grp) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[r] += g1s[r * 256 + tid]; acc2[r] += g1s[(16 + r) * 256 + tid]; }
      cs += g1s[32 * 256 + tid];
      cs2 += g1s[33 * 256 + tid];
    }
  }
  // column sums: lane (n, h) of waves 0, 1 holds the sum over its half of the samples; s_ones[0..63] of
  // delta, [64..127] of w (.) delta
  float* s_cs = smem - grp * 3 * WG_STAGE + 6 * WG_STAGE;     // [2 products][2 halves][64] (behind both groups' stages)
  float* s_ones = s_cs + 512;                 // [128]
  int* s_ticket = (int*)(s_ones + 128);
  if (sums && !grp) {
    s_cs[(lane >> 5) * 64 + wn * 32 + (lane & 31)] = cs;
    s_cs[128 + (lane >> 5) * 64 + wn * 32 + (lane & 31)] = cs2;
  }
  __syncthreads();
  if (ones && !grp && tid < 128) {
    const float* q = s_cs + (tid >> 6) * 128 + (tid & 63);
    s_ones[tid] = q[0] + q[64];
  }
  __syncthreads();
  const long long ldc = P.n_out;
  float* c1 = a.g1 + P.c_off;
  float* c2 = a.g2 + P.c_off;
  auto put = [&](float* dst, float v) { if (a.fresh) *dst = v; else *dst += v; };
  if (a.slices == 1) {                          // one slice: straight into the accumulators
    if (grp) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int m, n;
      wgrad_mn(tid, r, m, n);
      m += m0; n += n0;
      if (m < P.k_in && n < P.n_out) {
        put(c1 + (long long)m * ldc + n, acc[r]);
        put(c2 + (long long)m * ldc + n, acc2[r]);
      }
    }
    if (ones && tid < 128 && n0 + (tid & 63) < P.n_out)
      put((tid < 64 ? c1 : c2) + (long long)P.k_in * ldc + n0 + (tid & 63), s_ones[tid]);
    return;
  }
  // partial -> workspace: quad q of thread t at [q][t][4] (q < 4: first product, 4 .. 7: second, q = 8:
  // the ones rows); group 0 holds the sums
  float* part = a.ws + ((long long)tile * a.slices + slice) * WG_PART;
  if (!grp) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      wg_store_sc1(part + (q * 256 + tid) * 4, f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]});
      wg_store_sc1(part + ((q + 4) * 256 + tid) * 4, f32x4{acc2[4 * q], acc2[4 * q + 1], acc2[4 * q + 2], acc2[4 * q + 3]});
    }
    if (ones && tid < 32) wg_store_sc1(part + (WG_QUADS * 256 + tid) * 4, *(const f32x4*)(s_ones + 4 * tid));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores
  __syncthreads();
  WG_STAMP(3);
  if (threadIdx.x == 0) *s_ticket = __hip_atomic_fetch_add(a.tickets + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  WG_STAMP(4);
  if (*s_ticket != a.slices - 1) return;        // block-uniform: only the last arriver folds
  // Fold by all eight waves: group g takes product g + 1 (quads 4 g .. 4 g + 3), two quads = up to sixteen
  // 16-byte loads in flight per lane and round trip (the slices one at a time, quad after quad, were nine
  // dependent round trips: 7.5 us for 165 KB)
  const float* base = a.ws + (long long)tile * a.slices * WG_PART;
  float* cg = grp ? c2 : c1;
#pragma unroll
  for (int hq = 0; hq < 2; ++hq) {
    const int q0 = 2 * hq, q1 = q0 + 1;                       // quads of this group's product
    f32x4 v0, v1;
    wg_fold_quads2(base + ((4 * grp + q0) * 256 + tid) * 4, base + ((4 * grp + q1) * 256 + tid) * 4, WG_PART, a.slices, v0, v1);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int m, n;
      wgrad_mn(tid, 4 * (e < 4 ? q0 : q1) + (e & 3), m, n);
      m += m0; n += n0;
      if (m < P.k_in && n < P.n_out) put(cg + (long long)m * ldc + n, e < 4 ? v0[e] : v1[e & 3]);
    }
  }
  if (ones && !grp && tid < 32) {
    const f32x4 v = wg_fold_quad(base + (WG_QUADS * 256 + tid) * 4, WG_PART, a.slices);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = 4 * tid + e, n = n0 + (j & 63);
      if (n < P.n_out) put((j < 64 ? c1 : c2) + (long long)P.k_in * ldc + n, v[e]);
    }
  }
  if (threadIdx.x == 0) a.tickets[tile] = 0;    // for the next launch (stream-ordered after this one)
  WG_STAMP(5);
}

// The output layer's two sums (and its bias) from the per-workgroup partials of k_backprop16: output
// idx = s (H + 1) + k of fold block fb is taken by the four threads (idx, row group 0..3); a thread adds its
// quarter of the partial rows in ascending order, sixteen loads in flight, the four quarters are added as
// (0 + 1) + (2 + 3) -- a fixed order for a given batch size.
__device__ __forceinline__ void wgrad_fold_out(const WgradArgs& a, int fb, float* smem) {
  const int tid = threadIdx.x, o = tid & (WG_FOLD_OUT - 1), gq = tid >> 7;
  const int idx = fb * WG_FOLD_OUT + o, n_out = 2 * (a.out_H + 1);
  const bool live = idx < n_out;
  const int sidx = live ? idx / (a.out_H + 1) : 0, k = live ? idx - sidx * (a.out_H + 1) : 0;
  const int col = k < a.out_H ? k : a.out_ld - 4;                          // the bias slot: Hp
  const int per = (a.out_nwg + 3) >> 2, i0 = gq * per, i1 = min(a.out_nwg, i0 + per);
  const float* src = a.out_part + (long long)sidx * a.out_ld + col;
  const long long stride = 2LL * a.out_ld;
  float acc = 0.f;
  for (int i = i0; i < i1; i += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = src[(long long)min(i + u, i1 - 1) * stride];     // clamped: unconditional
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += i + u < i1 ? v[u] : 0.f;
  }
  smem[gq * WG_FOLD_OUT + o] = acc;
  __syncthreads();
  if (gq == 0 && live) {
    const float v = (smem[o] + smem[WG_FOLD_OUT + o]) + (smem[2 * WG_FOLD_OUT + o] + smem[3 * WG_FOLD_OUT + o]);
    float* dst = (sidx ? a.g2 : a.g1) + a.out_off + k;
    if (a.fresh) *dst = v; else *dst += v;
  }
}

__global__ __launch_bounds__(512) void k_wgrad(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wg_smem[];
  const int b = blockIdx.x;
  if (b < a.mfma_blocks) {
    const WgradBlock m = plan_wgrad_block(b, a.tiles, a.slices);
    if (m.slice < 0) return;
    int pi = 0;                                   // tile -> problem (block-uniform; a handful of problems)
    for (int i = 0; i < a.n_prob; ++i)
      if (m.tile >= a.prob[i].tile0) pi = i;      // tile0 ascends with the problem index
    const WgradProblem P = a.prob[pi];
    const int kbeg = m.slice * a.kchunk, klen = min(a.K, kbeg + a.kchunk) - kbeg;
    if ((klen & (WG_TK - 1)) == 0 && (long long)P.lda * 80 < (1LL << 31) && (long long)P.ldd * 80 < (1LL << 31))
      wgrad_tile<true>(a, P, m.slice, m.tile, wg_smem);
    else
      wgrad_tile<false>(a, P, m.slice, m.tile, wg_smem);
  } else if (b == a.mfma_blocks) {                // the scalar accumulators' block (always present in the grid)
    if (a.job.sc && threadIdx.x < 256)            // (waves 4 .. 7 retire: 256 threads do the scalars)
      scalar_accum_body(a.job, a.fresh != 0, (double*)wg_smem, (double*)wg_smem + 256);
  } else if (a.out_part) {
    wgrad_fold_out(a, b - a.mfma_blocks - 1, wg_smem);
  }
}

// more than half a CU's LDS: one workgroup per CU (a condition of the hand-over protocol above)
size_t wgrad_lds_bytes() {
  const size_t need = sizeof(float) * (6 * WG_STAGE + 512 + 128 + 4), one_per_cu = 81 * 1024;   // need: 114 KB
  return need > one_per_cu ? need : one_per_cu;
}

hipError_t launch_wgrad(hipStream_t s, const WgradLaunch& L) {
  WgradArgs a;
  memset((void*)&a, 0, sizeof(a));
  a.prob = (const WgradProblem*)L.dev_problems; a.n_prob = L.n_prob;
  a.tiles = L.tiles; a.slices = L.slices; a.kchunk = plan_wgrad_kchunk(L.K, L.slices);
  a.mfma_blocks = plan_wgrad_grid(L.tiles, L.slices);
  a.K = L.K; a.w = L.w; a.g1 = L.g1; a.g2 = L.g2; a.ws = L.ws; a.tickets = L.tickets; a.fresh = L.fresh ? 1 : 0;
  a.job = ScalarJob{L.sc_eloc, L.sc_ratio, L.sc_out, L.sc_B, L.sc_mode};

  a.out_part = L.out_part; a.out_nwg = L.out_nwg; a.out_H = L.out_H; a.out_ld = L.out_ld; a.out_off = L.out_off;
  const int n_fold = L.out_part ? plan_wgrad_fold_blocks(L.out_H) : 0;
  const int grid = a.mfma_blocks + ((L.sc_out || n_fold) ? 1 : 0) + n_fold;
  if (grid <= 0 || L.K <= 0) return hipSuccess;
  // The opt-in is per DEVICE and the library serves several devices per process (DeviceGuard): set it on
  // every launch like the other launchers (tail16.hpp, sweep16.hpp, conv_kernels.hpp), never cached.
  const size_t lds = wgrad_lds_bytes();
  {
    hipError_t e = hipFuncSetAttribute((const void*)k_wgrad, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
#ifdef VMC_WGRAD_STAMPS
  // diagnostic build: every 20th launch is stamped, read back synchronously and summarised on stderr
  static unsigned long long* d_st = nullptr;
  static int n_launch = 0;
  if (!d_st && hipMalloc((void**)&d_st, 4096 * 8 * sizeof(unsigned long long)) != hipSuccess) return hipErrorOutOfMemory;
  const bool probe = (++n_launch % 20) == 0 && grid <= 4096;
  if (probe) { if (hipMemsetAsync(d_st, 0, (size_t)grid * 64, s) != hipSuccess) return hipGetLastError(); a.stamps = d_st; }

#endif
  hipLaunchKernelGGL(k_wgrad, dim3(grid), dim3(512), lds, s, a);
#ifdef VMC_WGRAD_STAMPS
  if (probe) {
    std::vector<unsigned long long> h((size_t)grid * 8);
    if (hipStreamSynchronize(s) != hipSuccess ||
        hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return hipGetLastError();
    unsigned long long t0 = ~0ull, t_end = 0;
    for (int b = 0; b < a.mfma_blocks; ++b) if (h[(size_t)b * 8]) t0 = std::min(t0, h[(size_t)b * 8]);
    double sum[6] = {0, 0, 0, 0, 0, 0}, mx[6] = {0, 0, 0, 0, 0, 0};
    int n = 0, nf = 0;
    double cyc = 0.0, wall = 0.0;
    for (int b = 0; b < a.mfma_blocks; ++b) {
      const unsigned long long* q = &h[(size_t)b * 8];
      if (!q[0]) continue;
      ++n;
      for (int i = 0; i < 5; ++i) { const double v = (double)(q[i] - t0) * 0.01; sum[i] += v; mx[i] = std::max(mx[i], v); }
      if (q[5]) { ++nf; const double v = (double)(q[5] - t0) * 0.01; sum[5] += v; mx[5] = std::max(mx[5], v); }
      for (int i = 0; i < 6; ++i) t_end = std::max(t_end, q[i]);
      cyc += (double)(q[7] - q[6]); wall += (double)(q[2] - q[1]) * 0.01;
    }
    fprintf(stderr, "[wgrad stamps] loop: %.0f shader cycles (s_memtime) in %.2f us per workgroup = %.3f GHz; %.0f cycles per k-tile\n",
            cyc / n, wall / n, cyc / wall * 1e-3, cyc / n / (3.0 * ((a.kchunk / WG_TK + 2) / 3)));
    fprintf(stderr, "[wgrad stamps] blocks %d (folding %d) slices %d tiles %d | us since the first block started, mean / max: "
            "start %.2f/%.2f  loop %.2f/%.2f  loop end %.2f/%.2f  stores drained %.2f/%.2f  ticket %.2f/%.2f  fold end %.2f/%.2f | last stamp %.2f\n",
            n, nf, a.slices, a.tiles, sum[0] / n, mx[0], sum[1] / n, mx[1], sum[2] / n, mx[2], sum[3] / n, mx[3], sum[4] / n, mx[4],
            nf ? sum[5] / nf : 0.0, mx[5], (double)(t_end - t0) * 0.01);
  }
#endif
  return hipGetLastError();
}

size_t wgrad_problem_bytes() { return sizeof(WgradProblem); }

void wgrad_fill_problem(void* dst, int index, const float* A, long long lda, const float* D, long long ldd,
                        long long c_off, int k_in, int n_out, int tile0) {
  WgradProblem& p = ((WgradProblem*)dst)[index];
  p.A = A; p.lda = lda; p.D = D; p.ldd = ldd; p.c_off = c_off; p.k_in = k_in; p.n_out = n_out;
  p.tile0 = tile0; p.tiles_n = (n_out + WG_TN - 1) / WG_TN;
}

// The output tile of k_gemm128 / k_gemm_ring from its LDS image ct[128][ldt] (rows m0 .., columns n0 ..): thread
// tid of NT takes one column quad (N % 4 == 0: inside or outside as a whole) of every (NT / 32)-th row, in a
// ROLLED loop, so that the element epilogue exists four times in the code.
template <int NT>
__device__ __forceinline__ void gemm_tile_epilogue(const GemmArgs& g, const float* ct, int ldt, int m0, int n0, int tid) {
  const int cq = 4 * (tid & 31), n = n0 + cq;
  const bool in = n < g.N;
  if (g.epilogue == 10) {
    // row dot (the output layer folded into the last H x H layer): the 32 lanes of a row add their quads'
    // products in double -- e = 0 .. 3, then the butterfly 16, 8, 4, 2, 1: a fixed order -- and lane 0 stores the
    // tile's partial; nothing goes to C.  (A lane outside N adds zeros: every lane of a row stays in the shuffles.)
    f32x4 b = {0.f, 0.f, 0.f, 0.f}, w = {0.f, 0.f, 0.f, 0.f};
    if (in) { b = *(const f32x4*)(g.bias + n); w = *(const f32x4*)(g.dot_w + n); }
    double* const dst = g.dot_out + (long long)(n0 / 128) * g.M;
#pragma unroll 1
    for (int row = tid >> 5; row < 128; row += NT / 32) {
      const int m = m0 + row;
      if (m >= g.M) break;                        // (uniform over the 32 lanes of the row)
      const f32x4 v = *(const f32x4*)(ct + row * ldt + cq);
      double p = 0.0;
      if (g.act == VMC_ACT_RELU_) {
#pragma unroll
        for (int e = 0; e < 4; ++e) p += (double)vmc_act<VMC_ACT_RELU_>(v[e] + b[e]) * (double)w[e];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) p += (double)vmc_act_rt(g.act, v[e] + b[e]) * (double)w[e];
      }
      if (!in) p = 0.0;                           // (an activation of the padding need not be finite)
#pragma unroll
      for (int d = 16; d >= 1; d >>= 1) p += __shfl_xor(p, d);
      if ((tid & 31) == 0) dst[m] = p;
    }
    return;
  }
  if (!in) return;
#pragma unroll 1
  for (int row = tid >> 5; row < 128; row += NT / 32) {
    const int m = m0 + row;
    if (m >= g.M) break;
    const f32x4 v = *(const f32x4*)(ct + row * ldt + cq);
    if (g.epilogue == 1 && g.act == VMC_ACT_RELU_ && !g.dact_out) {   // the common case: relu(v + bias), one 16-byte store
      const f32x4 b = *(const f32x4*)(g.bias + n);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = vmc_act<VMC_ACT_RELU_>(v[e] + b[e]);
      *(f32x4*)(g.C + (long long)m * g.ldc + n) = o;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) gemm_epilogue(g, g.C, m, n + e, v[e]);
    }
  }
}

// ------------------------------------------------------------------- GEMM, large forward shape
// C[M,N] = epilogue(A[M,K] B[K,N]) for the shape the general path of wide.hip spends its time in: A = a
// block of materialised activation rows (k contiguous), B = a weight matrix [K][N] (n contiguous), M in
// the hundreds of thousands, K, N = the layer width.  128 x 128 x 16 tiles, 4 waves in a 2 x 2 grid, each
// 64 x 64 of the tile as 2 x 2 v_mfma_f32_32x32x2_f32 accumulators: one LDS operand read per MFMA (k_gemm:
// two), half the global bytes per flop of the 64 x 64 tile.  Two LDS stages: tile t + 1 travels into
// registers while tile t is multiplied and goes to the other stage behind it -- one barrier per tile.
// Out-of-range rows read a clamped row and are not stored; out-of-range k / n quads are zeroed when they
// are staged (not at the load: a select there makes the compiler wait for the load at once).
// Requirements (launch_gemm checks them, else k_gemm): sak == 1, sbn == 1, K % 4 == N % 4 == 0, rows of
// A and B 16-byte aligned, no split-K / dual / ones row.
// Block order: the N / 128 column tiles of one row tile run back to back on ONE XCD (block b -> XCD
// b % 8), so that a row block of A is fetched into one L2 once instead of into all eight.
#define G2_TM 128
#define G2_TN 128
#define G2_TK 16
#define G2_LD 132
#define GEMM_RING_MIN 160
// TN = 128.  (TN = 64 -- 128 x 64 tiles for grids the square tile does not fill the chip with -- was measured
// on the sampler's 4096 x 1024 x 1024 products: 2 % over k_gemm's 64 x 64 tiles, not kept as a launch path.)
// Epilogue (round 5): the tile goes to LDS in accumulator order and comes back as 16-byte row pieces in a ROLLED
// loop.  The unrolled form -- the element epilogue, a switch over epilogue kinds and activations, 64 times per
// thread -- was ~100 KB of instructions that every wave walked through once per tile (see k_gemm_ring).
// (An eight-wave form of this kernel, 2 x 4 waves of 64 x 32, for the sampler's one-tile-per-CU grids was measured
// in round 5: 95 us at 4096 x 1024 x 1024 against 111 with four waves and 86 for k_gemm; k_gemm_ring serves there.)
template <int TN>
__global__ __launch_bounds__(256, 2) void k_gemm128(GemmArgs g, int tiles_m, int tiles_n) {
  static_assert(TN == 128, "the LDS image of the output tile is [128][128]");
  constexpr int NJ = TN / 64;                        // 32-column accumulators per wave
  constexpr int BQ = TN / 4, BR = 256 / BQ, BP = G2_TK / BR;   // B tile: quads per k row, k rows per pass, passes
  // 64 KB: the two stages of both operands during the loop (33 KB), the output tile behind it
  __shared__ __attribute__((aligned(16))) float g2_lds[G2_TM * TN];
  float (*const As)[G2_TK][G2_LD] = (float (*)[G2_TK][G2_LD])g2_lds;
  float (*const Bs)[G2_TK][TN + 4] = (float (*)[G2_TK][TN + 4])(g2_lds + 2 * G2_TK * G2_LD);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: XCD x takes the row tiles x, x + 8, ..., each with all of its column tiles
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int tm = (j / tiles_n) * 8 + xcd, tn = j % tiles_n;
  if (tm >= tiles_m) return;                        // (whole workgroup; before any barrier)
  const int m0 = tm * G2_TM, n0 = tn * TN;
  const int T = (g.K + G2_TK - 1) / G2_TK;
  // this thread's two quads of the A tile (row am + 64 i, k quad ak) and of the B tile (k row bk + 8 i, n quad bn)
  const int am = tid >> 2, ak = 4 * (tid & 3);
  const int bk = tid / BQ, bn = 4 * (tid % BQ);
  const float* ap[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) ap[i] = g.A + (long long)min(m0 + am + 64 * i, g.M - 1) * g.sam;
  const bool b_in = n0 + bn < g.N;                  // N % 4 == 0: a quad is inside or outside as a whole
  const float* bp = g.B + (b_in ? n0 + bn : 0);
  f32x4 ra[2], rb[BP];
  const int sbk = (int)g.sbk;                       // K * sbk < 2^31 (gemm128_layout_ok)
  auto request = [&](int t) {
    const int k0 = min(t, T - 1) * G2_TK;
#pragma unroll
    for (int i = 0; i < 2; ++i) ra[i] = *(const f32x4*)(ap[i] + min(k0 + ak, g.K - 4));
#pragma unroll
    for (int i = 0; i < BP; ++i) rb[i] = *(const f32x4*)(bp + min(k0 + bk + BR * i, g.K - 1) * sbk);
    // (the loads stay HERE, a whole tile of MFMAs ahead of their use: without the barrier the compiler
    // sinks them behind the products, next to the stage that consumes them)
    __builtin_amdgcn_sched_barrier(0);
  };
  auto stage = [&](int t) {
    const int k0 = t * G2_TK, st = t & 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool a_ok = k0 + ak < g.K;              // K % 4 == 0
#pragma unroll
      for (int e = 0; e < 4; ++e) As[st][ak + e][am + 64 * i] = a_ok ? ra[i][e] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const bool b_ok = b_in && k0 + bk + BR * i < g.K;
      *(f32x4*)&Bs[st][bk + BR * i][bn] = b_ok ? rb[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
  request(0);
  stage(0);
  __syncthreads();
  const int l31 = lane & 31, hi = lane >> 5;
  for (int t = 0; t < T; ++t) {
    request(t + 1);                                 // (clamped: the last one re-reads tile T - 1, unused)
    const int st = t & 1;
    // operands of k-step s + 1 are read before the products of step s issue (one LDS round trip ahead)
    float av[2][2], bv[2][NJ];
    auto operands = [&](int kk, float (&a2)[2], float (&b2)[NJ]) {
#pragma unroll
      for (int i = 0; i < 2; ++i) a2[i] = As[st][kk + hi][wm * 64 + 32 * i + l31];
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) b2[jj] = Bs[st][kk + hi][wn * (TN / 2) + 32 * jj + l31];
    };
    operands(0, av[0], bv[0]);
#pragma unroll
    for (int ks = 0; ks < G2_TK / 2; ++ks) {
      if (ks + 1 < G2_TK / 2) operands(2 * (ks + 1), av[(ks + 1) & 1], bv[(ks + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
          acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks & 1][i], bv[ks & 1][jj], acc[i][jj], 0, 0, 0);
    }
    if (t + 1 < T) stage(t + 1);                    // uniform
    __syncthreads();
  }
  // (the barrier that ended the loop: every wave has read its last operands; the stages may be overwritten)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        g2_lds[(wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hi) * TN + wn * (TN / 2) + 32 * jj + l31] = acc[i][jj][r];
  __syncthreads();
  gemm_tile_epilogue<256>(g, g2_lds, TN, m0, n0, tid);
}

// ------------------------------------------------------------------- GEMM, one tile per CU (round 5)
// The general sampler's products (4096 chains x H x H, one per hidden layer and mc_step) are 256 tiles of
// 128 x 128: ONE round of the 256 CUs.  A kernel that stages through registers has one k-tile of look-ahead
// (k_gemm128: 16 k = 0.85 us of MFMAs at two waves per SIMD) -- less than a first touch of the activation
// rows takes -- and with one workgroup per CU nothing else runs while it waits: 111 us (4 waves) / 95 us
// (8 waves) against 86 for k_gemm's 1024 small tiles, whose four workgroups per CU cover for each other.
// k_gemm_ring keeps the large tile and fetches BOTH operands by LDS-DMA (global_load_lds_dwordx4: no VGPR
// round trip, nothing for a wave to wait on) into a ring of GR_RING stages of 32 k, three stages = 5 us ahead:
//   stage    A tile 128 rows x 32 k (16 KiB) + B tile 32 k x 128 n (16 KiB); eight waves fetch four 1 KiB
//            pieces each, one per item, under the MFMAs.
//   A image  a DMA instruction writes lane L's 16 bytes at [M0 + 16 L]; lane L fetches row 8 i + (L >> 3),
//            k quad (L & 7) ^ ((row >> 1) & 7): eight lanes cover one 128-byte line of a row, and the XOR puts the
//            quads an MFMA-layout read wants (sixteen consecutive rows, one k quad) on sixteen distinct bank
//            quads.  One ds_read_b128 = this lane's A operands of FOUR k-steps.
//   B image  [k][n] as in memory (a piece = two k rows); operands by ds_read2st64_b32 (two k rows per read).
//   k order  lane half h of an MFMA takes k = 4 (2 p + h) + e in item p, step e -- a permutation of the
//            stage's 32 k (the sum's order differs from k_gemm's: equal within rounding, not bit for bit).
//   waves    2 (m) x 4 (n), each 64 x 32 of the tile as two 32x32x2 accumulators sharing the B operand:
//            4 LDS instructions per 8 MFMAs.
//   protocol (every wave runs the same sequence; ONE s_barrier per stage = 2,048 matrix cycles per wave)
//     every item: issue one piece; wait for this item's operands, read the operands of the NEXT item (item 3:
//                 of stage s + 1, item 0); eight MFMAs.
//     item 3 of stage s, between that wait and those reads:  s_waitcnt vmcnt(8) -- my pieces of stage s + 1
//                 have landed, two stages' worth may be in flight -- and s_barrier: all of s + 1 has landed
//                 (it is about to be read), and every wave holds its last operands of stage s in registers,
//                 so the slot of s is free: the pieces of stage s + 4 go there, from this item on (item 3
//                 of s, items 0 .. 2 of s + 1).  A piece has 2.25 .. 3 stages (4 .. 7 us) to land.  (With
//                 the barrier at the stage boundary and the pieces of s + 3 issued in items 0 .. 3 of s the
//                 last piece had ONE stage, less than a miss to HBM takes under load: MFMA busy 0.61.)
//   The stages behind the last are re-fetches of the last (in bounds, never read): the counts stay uniform.
//   Every wave drains its DMA before it exits.  Rows beyond M / columns beyond N fetch a clamped row / quad
//   and are not stored.
// Requirements (gemm_ring_ok): the k_gemm128 layout, K % 32 == 0 (whole stages).
#define GR_TK 32
#define GR_RING 4
#define GR_ABYTES (G2_TM * GR_TK * 4)
#define GR_BBYTES (GR_TK * G2_TN * 4)
#define GR_LDS (GR_RING * (GR_ABYTES + GR_BBYTES))
#define GR_LDC (G2_TN + 4)                 // row stride of the output tile's LDS image (floats): 16-byte rows, banks spread
// DIAGNOSTIC builds only (-DVMC_GR_ABLATE=mask, tools/gemm_ring_ablate.sh; results are garbage, only the time is read):
// 1 no DMA in the stage loop, 2 no barrier, 4 no operand reads, 8 no MFMAs
#ifndef VMC_GR_ABLATE
#define VMC_GR_ABLATE 0
#endif
typedef float f32x2_gr __attribute__((ext_vector_type(2)));

// ONE 1 KiB piece: lane L's 16 bytes from [base + voff] to LDS [lds + 16 L].  M0 is saved and restored.
#define GR_DMA(BASE, VOFF, LDS)                                                                             \
  do {                                                                                                      \
    unsigned sv_;                                                                                           \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"     \
                 "s_mov_b32 m0, %0"                                                                         \
                 : "=&s"(sv_) : "s"(LDS), "v"(VOFF), "s"(BASE) : "memory");                                 \
  } while (0)
// ONE statement per item: wait for the operands of the item about to be multiplied (set S, issued by the
// previous statement), then issue the reads of the item behind it (ring slot SLOT, item P) into the other
// set T.  Both sets are in/out operands: the MFMAs of this item follow the statement (they read its
// outputs), the next statement follows them (it redefines set S) -- tail_split.hip, LDS_STEP.  The
// accumulators are in/out operands too (the statement does not touch them): without that tie the compiler
// renames set S with register copies and sinks the MFMAs of an item behind the NEXT statement's wait.
#if VMC_GR_ABLATE & 4
#define GR_READS(T, SLOT, P) "; %[ta0] %[pa0] %[oa] %[ta1] %[pa1] %[tb0] %[pb] %[ob] %[tb1]"
#else
#define GR_READS(T, SLOT, P)                                                                                \
  "ds_read_b128 %[ta0], %[pa0] offset:%[oa]\n\tds_read_b128 %[ta1], %[pa1] offset:%[oa]\n\t"               \
  "ds_read2st64_b32 %[tb0], %[pb] offset0:%[ob] offset1:%[ob]+2\n\t"                                        \
  "ds_read2st64_b32 %[tb1], %[pb] offset0:%[ob]+4 offset1:%[ob]+6"
#endif
#if VMC_GR_ABLATE & 2
#define GR_BARRIER "s_nop 0"
#else
#define GR_BARRIER "s_barrier"
#endif
#define GR_STEP_(WAIT, S, T, SLOT, P)                                                                       \
  asm volatile(WAIT "\n\t" GR_READS(T, SLOT, P)                                                              \
               : [sa0] "+v"(a0[S]), [sa1] "+v"(a1[S]), [sb0] "+v"(b01[S]), [sb1] "+v"(b23[S]),               \
                 [ta0] "+v"(a0[T]), [ta1] "+v"(a1[T]), [tb0] "+v"(b01[T]), [tb1] "+v"(b23[T]),               \
                 "+v"(acc0), "+v"(acc1)                                                                     \
               : [pa0] "v"(adA[0][P]), [pa1] "v"(adA[1][P]), [pb] "v"(adB),                                 \
                 [oa] "n"((SLOT) * GR_ABYTES), [ob] "n"((SLOT) * (GR_BBYTES / 256) + (P) * 16)              \
               : "memory")
#define GR_STEP(S, T, SLOT, P) GR_STEP_("s_waitcnt lgkmcnt(0)", S, T, SLOT, P)
// item 3: the stage hand-over (see the protocol above) sits between the wait and the reads
#define GR_STEP_NEXT(S, T, SLOT) GR_STEP_("s_waitcnt vmcnt(8) lgkmcnt(0)\n\t" GR_BARRIER, S, T, SLOT, 0)

// WHOLE: K is a multiple of 128 -- whole turns of the ring, no stage is skipped (no branch around the MFMAs)
// CONVA: the A pieces gather a periodic convolution's input on the fly (GemmArgs.conv_a): lane L of a piece still takes row
// 8 i + (L >> 3), k quad (L & 7) ^ ((row >> 1) & 7) of the stage -- a stage is 32 channels of ONE tap (ca_F % 32 == 0) -- but
// the row is a lattice position and the address that of the tap's periodic neighbour: no im2col matrix is written or read.
template <bool WHOLE, bool CONVA = false>
__global__ __launch_bounds__(512) void k_gemm_ring(GemmArgs g, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char gr_lds[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  // XCD-aware tile order (k_gemm128): XCD x takes the row tiles x, x + 8, ..., each with all of its column tiles
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int tm = (jb / tiles_n) * 8 + xcd, tn = jb % tiles_n;
  if (tm >= tiles_m) return;                        // (whole workgroup; before any barrier or DMA)
  const int m0 = tm * G2_TM, n0 = tn * G2_TN;
  const int T = g.K / GR_TK;
  const int sam = (int)g.sam, sbk = (int)g.sbk;     // 128 sam, 4 sbk + N < 2^29 (gemm_ring_ok)
  const unsigned lds0 = (unsigned)(size_t)gr_lds;

  // ---- this wave's four pieces of a stage: A pieces 2 wave, 2 wave + 1 (eight rows each), B pieces likewise (two k rows each)
  unsigned voffA[2], voffB[2];
  int ca1[2] = {0, 0}, ca2[2] = {0, 0};             // CONVA: lattice coordinates of this lane's two rows
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int r = 8 * (2 * wave + u) + (lane >> 3);
    const int rc = min(m0 + r, g.M - 1) - m0;
    const int q = (lane & 7) ^ ((r >> 1) & 7);
    if (CONVA) {                                    // voffA: byte offset of (its row configuration, site 0, quad q)
      const int m = m0 + rc, rr = m / g.ca_N, n = m - rr * g.ca_N;
      ca1[u] = n / g.ca_D2; ca2[u] = n - ca1[u] * g.ca_D2;
      voffA[u] = (unsigned)(rr * g.ca_N * g.ca_Fp + 4 * q) * 4u;
    } else
    voffA[u] = (unsigned)(rc * sam + 4 * q) * 4u;
    const int nq = min(n0 + 4 * (lane & 31), g.N - 4);
    voffB[u] = (unsigned)((2 * u + (lane >> 5)) * sbk + nq) * 4u;
  }
  const float* const abase = g.A + (long long)m0 * g.sam;
  const float* const bbase = g.B + (long long)(4 * wave) * g.sbk;
  const unsigned ldsA = lds0 + (2 * wave) * 1024, ldsB = lds0 + GR_RING * GR_ABYTES + (2 * wave) * 1024;
  auto sgpr_ptr = [](const float* p) {              // (tail_split.hip: an "s" operand must really be in SGPRs)
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi32 = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const float*)(((unsigned long long)hi32 << 32) | lo);
  };
  // piece p (0, 1: A; 2, 3: B) of stage s into ring slot `slot`
  auto piece = [&](int p, int s, int slot) {
    if ((VMC_GR_ABLATE & 1) && s >= GR_RING) return;
    const int k0 = min(s, T - 1) * GR_TK;
    if (p < 2 && CONVA) {
      // the stage's tap (uniform) and, per lane, the periodic neighbour of its row's site under that tap
      const int t = k0 / g.ca_F, c0 = k0 - t * g.ca_F, d1 = t / g.ca_KW, d2 = t - d1 * g.ca_KW;
      int s1 = ca1[p] + d1 - g.ca_lo, s2 = ca2[p] + d2 - g.ca_lo2;
      s1 += s1 < 0 ? g.ca_D1 : 0; s1 -= s1 >= g.ca_D1 ? g.ca_D1 : 0;
      s2 += s2 < 0 ? g.ca_D2 : 0; s2 -= s2 >= g.ca_D2 ? g.ca_D2 : 0;
      const unsigned vo = voffA[p] + (unsigned)((s1 * g.ca_D2 + s2) * g.ca_Fp + c0) * 4u;
      const float* b_ = sgpr_ptr(g.A);
      const unsigned l_ = __builtin_amdgcn_readfirstlane(ldsA + slot * GR_ABYTES + p * 1024);
      GR_DMA(b_, vo, l_);
    } else if (p < 2) {
      const float* b_ = sgpr_ptr(abase + k0);
      const unsigned l_ = __builtin_amdgcn_readfirstlane(ldsA + slot * GR_ABYTES + p * 1024);
      GR_DMA(b_, voffA[p], l_);
    } else {
      const float* b_ = sgpr_ptr(bbase + (long long)k0 * sbk);
      const unsigned l_ = __builtin_amdgcn_readfirstlane(ldsB + slot * GR_BBYTES + (p - 2) * 1024);
      GR_DMA(b_, voffB[p - 2], l_);
    }
  };

  // ---- operand addresses: A row m = 64 wm + 32 i + l31, k quad 2 p + hi (swizzled as it was fetched); B row hi, column n
  unsigned adA[2][4], adB;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int m = wm * 64 + 32 * i + l31;
      adA[i][p] = lds0 + (unsigned)(m * 8 + ((2 * p + hi) ^ ((m >> 1) & 7))) * 16u;
    }
  adB = lds0 + GR_RING * GR_ABYTES + (unsigned)(hi * 4 * G2_TN + wn * 32 + l31) * 4u;

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  f32x4 a0[2], a1[2];
  f32x2_gr b01[2], b23[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { a0[i] = f32x4{0.f, 0.f, 0.f, 0.f}; a1[i] = a0[i]; b01[i] = f32x2_gr{0.f, 0.f}; b23[i] = b01[i]; }

#pragma unroll
  for (int s = 0; s < GR_RING - 1; ++s) {
#pragma unroll
    for (int p = 0; p < 4; ++p) piece(p, s, s);
  }
  asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier\n\t" GR_READS(0, 0, 0)     // stage 0 has landed: mine, everybody's
               : [ta0] "+v"(a0[0]), [ta1] "+v"(a1[0]), [tb0] "+v"(b01[0]), [tb1] "+v"(b23[0])
               : [pa0] "v"(adA[0][0]), [pa1] "v"(adA[1][0]), [pb] "v"(adB), [oa] "n"(0), [ob] "n"(0) : "memory");
  piece(0, GR_RING - 1, GR_RING - 1);

#define GR_MFMA8(S)                                                                          \
  do {                                                                                       \
    if ((VMC_GR_ABLATE & 8) || (!WHOLE && s_ >= T)) break;   /* (a stage behind the last: the turn is completed, nothing is added) */ \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[S][0], b01[S][0], acc0, 0, 0, 0);         \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[S][0], b01[S][0], acc1, 0, 0, 0);         \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[S][1], b01[S][1], acc0, 0, 0, 0);         \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[S][1], b01[S][1], acc1, 0, 0, 0);         \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[S][2], b23[S][0], acc0, 0, 0, 0);         \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[S][2], b23[S][0], acc1, 0, 0, 0);         \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[S][3], b23[S][1], acc0, 0, 0, 0);         \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[S][3], b23[S][1], acc1, 0, 0, 0);         \
  } while (0)
  // one stage in ring slot SLOT (a compile-time constant: the LDS offsets are instruction immediates)
#define GR_STAGE(SLOT)                                                                       \
  do {                                                                                       \
    const int s_ = st + (SLOT);                                                              \
    piece(1, s_ + 3, ((SLOT) + 3) & 3); GR_STEP(0, 1, SLOT, 1); GR_MFMA8(0);                 \
    piece(2, s_ + 3, ((SLOT) + 3) & 3); GR_STEP(1, 0, SLOT, 2); GR_MFMA8(1);                 \
    piece(3, s_ + 3, ((SLOT) + 3) & 3); GR_STEP(0, 1, SLOT, 3); GR_MFMA8(0);                 \
    GR_STEP_NEXT(1, 0, ((SLOT) + 1) & 3); piece(0, s_ + 4, SLOT); GR_MFMA8(1);               \
  } while (0)
  for (int st = 0; st < T; st += GR_RING) {         // whole turns of the ring (the slot is an instruction immediate)
    GR_STAGE(0); GR_STAGE(1); GR_STAGE(2); GR_STAGE(3);
  }
  // the read issued by the last item (a stage never multiplied) and this wave's re-fetches land before it leaves
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
               : "+v"(a0[0]), "+v"(a1[0]), "+v"(b01[0]), "+v"(b23[0]) :: "memory");
#undef GR_STAGE
#undef GR_MFMA8

  // Epilogue through LDS (the ring is drained: every wave waited for its DMA and holds no operand in flight): the
  // tile goes to LDS in accumulator order and comes back as 16-byte row pieces in a ROLLED loop
  // (gemm_tile_epilogue) -- the element epilogue, a switch over epilogue kinds and activations, exists four
  // times in the code.  Unrolled, 32 times per thread, it was ~100 KB of instructions that every wave walked
  // through once, one instruction-cache miss after the other: 22 us of an 83 us launch were outside the stage
  // loop (mask 15 of tools/gemm_ring_ablate.sh), 9 us with this form.
  __syncthreads();
  float* const ct = (float*)gr_lds;                 // [128][GR_LDC]
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const f32x16& v = i ? acc1 : acc0;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      ct[(wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hi) * GR_LDC + wn * 32 + l31] = v[r];
  }
  __syncthreads();
  gemm_tile_epilogue<512>(g, ct, GR_LDC, m0, n0, tid);
}

static bool gemm128_layout_ok(const GemmArgs& g);
static bool gemm_ring_ok(const GemmArgs& g) {
  return gemm128_layout_ok(g) && g.N >= 64 && g.K % GR_TK == 0 && g.K >= 4 * GR_TK &&
         g.sam * 128 < (1LL << 29) && g.sbk * 4 + g.N < (1LL << 29);
}

// CGS_VMC_GEMM128=0: the 64 x 64 kernel for every shape (A/B measurements); =2: k_gemm128 wherever its operand
// layout allows, whatever the grid size (tests: small shapes with ragged edges); =5: k_gemm_ring wherever it
// applies; =4: the round-4 rule (k_gemm_ring never taken).
// Read at every launch so that a test can compare the tilings in one process.
static int gemm128_mode() { const char* e = getenv("CGS_VMC_GEMM128"); return e ? atoi(e) : 1; }
static bool gemm128_layout_ok(const GemmArgs& g) {
  return g.sak == 1 && g.sbn == 1 && g.splitk <= 1 && !g.dual && !g.ones_row && !g.kscale &&
         g.N >= 64 && g.K >= 4 * G2_TK && g.K % 4 == 0 && g.N % 4 == 0 &&
         g.sam % 4 == 0 && g.sbk % 4 == 0 && ((size_t)g.A & 15) == 0 && ((size_t)g.B & 15) == 0 &&
         (long long)g.K * g.sbk < (1LL << 31);
}
// 0: k_gemm; 1: k_gemm128; 2: k_gemm_ring
static int gemm_tiling(const GemmArgs& g) {
  const int mode = gemm128_mode();
  // (64 <= N < 128: ONE column tile with its upper columns clamped and masked -- the general convolution path's 65 .. 127
  // filters; the 64 x 64 kernel would read the im2col rows twice.  CGS_VMC_GEMM_NARROW=0: as before round 5's last change)
  const char* narrow_env = getenv("CGS_VMC_GEMM_NARROW");       // read per launch (A/B tests in one process), as CGS_VMC_GEMM128
  const bool narrow = !(narrow_env && atoi(narrow_env) == 0);
  if (mode == 0 || !gemm128_layout_ok(g) || g.N < (narrow ? 64 : G2_TN)) return 0;
  if (mode == 2) return 1;
  if (mode == 5) return gemm_ring_ok(g) ? 2 : 0;
  const long long tiles = (long long)((g.M + G2_TM - 1) / G2_TM) * ((g.N + G2_TN - 1) / G2_TN);
  // k_gemm_ring (one workgroup per CU) from GEMM_RING_MIN tiles on -- below that the 64 x 64 tiles' larger grid
  // fills more CUs; k_gemm128 where the ring does not apply (K % 32 != 0), from four workgroups per CU on
  if (mode != 4 && tiles >= GEMM_RING_MIN && gemm_ring_ok(g)) return 2;
  return tiles >= 1024 ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_gemm_reduce(GemmArgs g) {
  const long long mn = (long long)g.M * g.N;
  const int nd = g.dual ? 2 : 1;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < mn * nd;
       i += (long long)gridDim.x * 256) {
    const int d = (int)(i / mn);
    const long long e = i % mn;
    double v = 0.0;                  // the slices are folded in z order, in double
    for (int z = 0; z < g.splitk; ++z) v += (double)g.workspace[((long long)z * nd + d) * mn + e];
    gemm_epilogue(g, d ? g.C2 : g.C, (int)(e / g.N), (int)(e % g.N), (float)v);
  }
}

bool gemm_rowdot_ok(const GemmArgs& g) { return g.M > 0 && g.N > 0 && gemm_tiling(g) != 0; }
// the implicit-gather A operand: the ring kernel only; whole stages inside one tap; 32-bit byte offsets into the map
bool gemm_conv_a_ok(const GemmArgs& g) {
  if (!g.conv_a || g.M <= 0 || g.ca_F % GR_TK != 0 || g.ca_F <= 0 || g.K % g.ca_F != 0 || ((size_t)g.A & 15) || g.ca_Fp % 4 != 0) return false;
  if ((long long)g.M * g.ca_Fp * 4 >= (1LL << 32)) return false;
  GemmArgs t = g;
  t.sam = g.K; t.sak = 1;                            // (the layout checks of the plain form, with a matrix-like stride)
  return gemm_tiling(t) == 2;
}

hipError_t launch_gemm(hipStream_t s, const GemmArgs& g_in) {
  if (g_in.M <= 0 || g_in.N <= 0) return hipSuccess;
  GemmArgs g = g_in;
  if (g.conv_a) {
    if (!gemm_conv_a_ok(g)) return hipErrorInvalidValue;
    g.sam = g.K; g.sak = 1;                          // (what the layout checks of gemm_tiling read; the kernel does not)
  }
  if (const int tiling = gemm_tiling(g)) {
    const int tiles_m = (g.M + G2_TM - 1) / G2_TM, tiles_n = (g.N + G2_TN - 1) / G2_TN;
    const int blocks = ((tiles_m + 7) / 8) * 8 * tiles_n;      // whole rounds of the eight XCDs
    if (tiling == 2) {
      // (the opt-in is per device: set on every launch, as launch_wgrad does)
      const bool whole = g.K % (GR_RING * GR_TK) == 0;
      const void* f = g.conv_a ? (whole ? (const void*)k_gemm_ring<true, true> : (const void*)k_gemm_ring<false, true>)
                               : (whole ? (const void*)k_gemm_ring<true> : (const void*)k_gemm_ring<false>);
      hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, GR_LDS);
      if (e != hipSuccess) return e;
      if (g.conv_a) {
        if (whole) hipLaunchKernelGGL((k_gemm_ring<true, true>), dim3(blocks), dim3(512), GR_LDS, s, g, tiles_m, tiles_n);
        else hipLaunchKernelGGL((k_gemm_ring<false, true>), dim3(blocks), dim3(512), GR_LDS, s, g, tiles_m, tiles_n);
      } else if (whole) hipLaunchKernelGGL(k_gemm_ring<true>, dim3(blocks), dim3(512), GR_LDS, s, g, tiles_m, tiles_n);
      else hipLaunchKernelGGL(k_gemm_ring<false>, dim3(blocks), dim3(512), GR_LDS, s, g, tiles_m, tiles_n);
    } else {
      hipLaunchKernelGGL(k_gemm128<G2_TN>, dim3(blocks), dim3(256), 0, s, g, tiles_m, tiles_n);
    }
    return hipGetLastError();
  }
  if (g.epilogue == 10) return hipErrorInvalidValue;           // only the tile kernels above (gemm_rowdot_ok)
  const int m_rows = g.ones_row ? g.M - 1 : g.M;
  const dim3 grid((g.N + GT - 1) / GT, (m_rows + GT - 1) / GT, g.splitk);
  if (g.dual) hipLaunchKernelGGL((k_gemm<true>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((k_gemm<false>), grid, dim3(256), 0, s, g);
  if (g.splitk > 1) {
    const long long total = (long long)g.M * g.N * (g.dual ? 2 : 1);
    const int blocks = (int)min((total + 255) / 256, (long long)2048);
    hipLaunchKernelGGL(k_gemm_reduce, dim3(blocks), dim3(256), 0, s, g);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------- element-wise
// a = f(z) (and, where given, f'(z)): the first activation of the gradient path's own forward
__global__ void k_act_copy(const float* __restrict__ z, float* __restrict__ a,
                           float* __restrict__ dact, long long n, int act) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float v = vmc_act_rt(act, z[i]);
    a[i] = v;
    if (dact) dact[i] = vmc_dact_rt(act, z[i], v);
  }
}

hipError_t launch_act_copy(hipStream_t s, const float* z, float* a, float* dact, long long n, int act) {
  const int blocks = (int)min((n + 255) / 256, (long long)4096);
  hipLaunchKernelGGL(k_act_copy, dim3(blocks), dim3(256), 0, s, z, a, dact, n, act);
  return hipGetLastError();
}

// per-sample factors of a non-exp output activation g: psi = g(x), oscale = g'(x) / g(x)
__global__ void k_out_scale(const float* __restrict__ x, float* __restrict__ oscale, int B, int oact) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) oscale[i] = vmc_out_dlog(oact, x[i]);
}

hipError_t launch_out_scale(hipStream_t s, const float* x, float* oscale, int B, int oact) {
  hipLaunchKernelGGL(k_out_scale, dim3((B + 255) / 256), dim3(256), 0, s, x, oscale, B, oact);
  return hipGetLastError();
}

__global__ void k_tanh_copy(const float* __restrict__ z, float* __restrict__ a, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    a[i] = tanhf(z[i]);
}

hipError_t launch_tanh_copy(hipStream_t s, const float* z, float* a, long long n) {
  const int blocks = (int)min((n + 255) / 256, (long long)4096);
  hipLaunchKernelGGL(k_tanh_copy, dim3(blocks), dim3(256), 0, s, z, a, n);
  return hipGetLastError();
}

__global__ void k_scale_one(float* x, float f) { x[0] *= f; }

hipError_t launch_scale_one(hipStream_t s, float* x, float f) {
  hipLaunchKernelGGL(k_scale_one, dim3(1), dim3(1), 0, s, x, f);
  return hipGetLastError();
}

__global__ void k_fill(float* __restrict__ x, float v, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    x[i] = v;
}

hipError_t launch_fill(hipStream_t s, float* x, float v, long long n) {
  if (n <= 0) return hipSuccess;
  const int blocks = (int)min((n + 255) / 256, (long long)4096);
  hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, s, x, v, n);
  return hipGetLastError();
}

// tf.metrics.mean updates (training.py:555, 689-690) + mean_tensor count (550-553):
// scalars = [e_total, e_count, r_total, r_count, g_count]
__global__ __launch_bounds__(1024) void k_scalar_accum(const float* __restrict__ eloc,
                                                       const float* __restrict__ ratio, int B,
                                                       float* __restrict__ sc, int mode, int fresh) {
  __shared__ double se[1024];
  __shared__ double sr[1024];
  double e = 0.0, r = 0.0;
  for (int i = threadIdx.x; i < B; i += 1024) {
    e += (double)eloc[i];
    if (ratio) r += (double)ratio[i];
  }
  se[threadIdx.x] = e; sr[threadIdx.x] = r;
  __syncthreads();
  for (int d = 512; d >= 1; d >>= 1) {
    if (threadIdx.x < d) { se[threadIdx.x] += se[threadIdx.x + d]; sr[threadIdx.x] += sr[threadIdx.x + d]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (fresh) {   // first accumulate after reset_gradients: the scalars hold no sum yet
#pragma unroll
      for (int i = 0; i < 8; ++i) sc[i] = 0.f;
    }
    sc[0] += (float)se[0];
    sc[1] += (float)B;
    if (mode == 1) { sc[2] += (float)sr[0]; sc[3] += (float)B; }
    sc[4] += 1.f;
  }
}

hipError_t launch_scalar_accum(hipStream_t s, const float* eloc, const float* ratio, int B,
                               float* acc_scalars, int mode, bool fresh) {
  hipLaunchKernelGGL(k_scalar_accum, dim3(1), dim3(1024), 0, s, eloc, ratio, B, acc_scalars,
                     mode, fresh ? 1 : 0);
  return hipGetLastError();
}

// ratio_b = (psi_w - beta H psi_w)/psi  (training.py:665-672)
//         = exp(logit_w - logit_psi + shift_psi - shift_w) * (1 - beta * E_loc^w)
// (a non-exp output activation g has no shift: psi_w / psi = g(x_w) / g(x_psi))
__global__ void k_itswo_ratio(const float* __restrict__ lp, const float* __restrict__ lw,
                              const float* __restrict__ ew, float log_factor, float beta, int B,
                              float* __restrict__ ratio, int oact) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const float amp = oact == VMC_ACT_EXP_ ? expf(lw[i] - lp[i] + log_factor)
                                         : vmc_act_rt(oact, lw[i]) / vmc_act_rt(oact, lp[i]);
  ratio[i] = amp * (1.f - beta * ew[i]);
}

hipError_t launch_itswo_ratio(hipStream_t s, const float* logit_psi, const float* logit_omega,
                              const float* eloc_omega, float log_factor, float beta, int B,
                              float* ratio, int oact) {
  hipLaunchKernelGGL(k_itswo_ratio, dim3((B + 255) / 256), dim3(256), 0, s, logit_psi,
                     logit_omega, eloc_omega, log_factor, beta, B, ratio, oact);
  return hipGetLastError();
}

// gradient formula (training.py:560-564 / 697-699) + TF1 Adam (training.py:84-91)
//   acc = [g1 (P) | g2 (P) | e_total e_count r_total r_count g_count ...]
__global__ void k_adam(float* __restrict__ theta, float* __restrict__ m, float* __restrict__ v,
                       const float* __restrict__ acc, int P, int mode, float lr_t, float b1,
                       float b2, float eps, float* __restrict__ grad_out, int apply) {
  const float* sc = acc + 2LL * P;
  const float gc = sc[4];
  const float mean_e = sc[0] / sc[1];
  const float mean_r = sc[2] / sc[3];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
    const float g1 = acc[i] / gc, g2 = acc[P + i] / gc;
    const float g = mode == 0 ? g2 - mean_e * g1 : g1 - g2 / mean_r;
    if (grad_out) grad_out[i] = g;
    if (apply) {
      const float mi = m[i] + (g - m[i]) * (1.f - b1);
      const float vi = v[i] + (g * g - v[i]) * (1.f - b2);
      m[i] = mi; v[i] = vi;
      theta[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
  }
}

hipError_t launch_adam(hipStream_t s, float* theta, float* m, float* v, const float* acc, int P,
                       int mode, float lr_t, float b1, float b2, float eps, float* grad_out) {
  const int blocks = min((P + 255) / 256, 2048);
  hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, s, theta, m, v, acc, P, mode, lr_t, b1,
                     b2, eps, grad_out, theta != nullptr ? 1 : 0);
  return hipGetLastError();
}
