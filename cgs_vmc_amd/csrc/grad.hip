// Gradient-accumulator path: the two tf.gradients calls of training.py:545-547 / 674-679
// (sum_b O_k(b) and sum_b w_b O_k(b), O_k = d logit / d theta_k) as an explicit
// forward / back-prop (k_backprop16, mlp.hip) / weight-gradient GEMM chain on fp32 MFMA, plus the accumulator,
// ratio and Adam element-wise kernels.  All reductions are fixed-order (no float atomics).
#include "common.hpp"

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
    case 8: *c = *c + v + g.bias[n]; break;
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
  float cs = 0.f, cs2 = 0.f;

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
        cs += Bs[kk][tid];
        if (DUAL) cs2 += Bs2[kk][tid];
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
      ws[(long long)m * g.N + nn] = cs;
      if (DUAL) ws[mn + (long long)m * g.N + nn] = cs2;
    } else {
      gemm_epilogue(g, g.C, m, nn, cs);
      if (DUAL) gemm_epilogue(g, g.C2, m, nn, cs2);
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
// ALL layers run as ONE launch of three kinds of workgroup (block-uniform dispatch on the block id):
//  * MFMA tiles: a 128 x 64 output tile of one layer over one K slice of the samples.  4 waves as 2 x 2,
//    each 64 x 32 = two 32x32x2 accumulators per product (four MFMAs per four LDS operand reads); the
//    operands of k-tile t+1 travel global -> registers while tile t is multiplied from LDS and go to the
//    OTHER LDS buffer afterwards: one barrier per 32 samples.  Slices are sized so that tiles x slices
//    fills the CUs once (plan_wgrad_slices); slice s lives on XCD s % 8, whose L2 then fetches that
//    slice's operand rows once (plan_wgrad_block).  The partial tile goes to the workspace in
//    accumulator order (coalesced), the workgroup takes a ticket, and the LAST one to arrive folds the
//    slices 0 .. S-1 of its tile in that fixed order into the accumulators -- no reduction launch, no
//    float atomics, the same bits whoever arrives last.
//  * column sums: the N = 1 problems (output layer: d logit / d w_out = a_L; RBM onsite layer) are
//    [a | 1]^T s with a per-sample scalar s -- VALU work, 64 columns per workgroup over all samples.
//  * one workgroup for the scalar accumulators (sum E, counts) of the same accumulate call.
struct WgradProblem {
  const float* A; long long lda;      // A(b, m) = A[b * lda + m], m < k_in (m == k_in: the ones row)
  const float* D; long long ldd;      // delta(b, n) = D[b * ldd + n], n < n_out
  long long c_off;                    // offset of C[(k_in + 1)][n_out] in g1 / g2 (theta layout: w then b)
  int k_in, n_out;
  int tile0, tiles_n;                 // first MFMA tile of this problem, tiles along n (n_out > 1)
  int col0;                           // first column-sum block of this problem (n_out == 1)
};

struct WgradArgs {
  const WgradProblem* prob;           // device table
  int n_prob;
  int tiles, slices, kchunk;          // MFMA tiles over all problems, K slices, samples per slice
  int mfma_blocks, col_blocks;        // block id ranges: [0, mfma_blocks) tiles, then column sums, then the scalars
  int K;                              // samples
  const float* w;                     // [K] per-sample weight of the second sum
  float* g1; float* g2;
  float* ws;                          // [tiles][slices][2][WG_TM * WG_TN + WG_TN] partial tiles + ones rows
  int* tickets;                       // [tiles], zero between launches
  int fresh;                          // 1: the accumulators hold no sum yet (store instead of add)
  ScalarJob job;
};

#define WG_LDA (WG_TM + 4)
#define WG_LDB (WG_TN + 4)
#define WG_STAGE (WG_TK * WG_LDA + 2 * WG_TK * WG_LDB)      // floats of one LDS stage: A, B, w (.) B
#define WG_PART (WG_TM * WG_TN + WG_TN)                      // floats of one partial (tile + ones row)

struct WgradRegs { f32x4 a[4]; f32x4 b[2]; float w[2]; };

// operands of the k-tile starting at sample k0 (all loads unconditional and clamped: an out-of-range
// row or column is read somewhere valid and zeroed, so that vmcnt is counted exactly)
__device__ __forceinline__ void wgrad_load(const WgradProblem& P, const float* __restrict__ w, int m0, int n0,
                                           int k0, int kend, int tid, bool vec_a, bool vec_d, WgradRegs& r) {
  const int am = m0 + 4 * (tid & 31);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = k0 + (tid >> 5) + 8 * i;
    const int kc = k < kend ? k : kend - 1;
    const float* p = P.A + (long long)kc * P.lda;
    f32x4 v;
    if (vec_a && am + 3 < P.k_in) {
      v = *(const f32x4*)(p + am);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = am + j < P.k_in ? p[am + j] : 0.f;
    }
    r.a[i] = k < kend ? v : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int bn = n0 + 4 * (tid & 15);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int k = k0 + (tid >> 4) + 16 * i;
    const int kc = k < kend ? k : kend - 1;
    const float* p = P.D + (long long)kc * P.ldd;
    f32x4 v;
    if (vec_d && bn + 3 < P.n_out) {
      v = *(const f32x4*)(p + bn);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = bn + j < P.n_out ? p[bn + j] : 0.f;
    }
    r.b[i] = k < kend ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    r.w[i] = w[kc];
  }
}

__device__ __forceinline__ void wgrad_stage(float* st, int tid, const WgradRegs& r) {
  float* As = st;
  float* Bs = st + WG_TK * WG_LDA;
  float* Bs2 = Bs + WG_TK * WG_LDB;
#pragma unroll
  for (int i = 0; i < 4; ++i) *(f32x4*)(As + ((tid >> 5) + 8 * i) * WG_LDA + 4 * (tid & 31)) = r.a[i];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int o = ((tid >> 4) + 16 * i) * WG_LDB + 4 * (tid & 15);
    *(f32x4*)(Bs + o) = r.b[i];
    *(f32x4*)(Bs2 + o) = r.b[i] * r.w[i];
  }
}

// element i (0 .. 31) of a thread's accumulators: i = 16 blk + r -> (m, n) inside the tile
__device__ __forceinline__ void wgrad_mn(int tid, int i, int& m, int& n) {
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int blk = i >> 4, r = i & 15;
  m = wm * 64 + blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
  n = wn * 32 + (lane & 31);
}

__device__ __forceinline__ void wgrad_tile(const WgradArgs& a, int slice, int tile, float* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  // tile -> problem (block-uniform; a handful of problems)
  int pi = 0;
  for (int i = 0; i < a.n_prob; ++i)
    if (a.prob[i].n_out > 1 && tile >= a.prob[i].tile0) pi = i;      // tile0 ascends with the problem index
  const WgradProblem P = a.prob[pi];
  const int lt = tile - P.tile0, tm = lt / P.tiles_n, tn = lt % P.tiles_n;
  const int m0 = tm * WG_TM, n0 = tn * WG_TN;
  const int kbeg = slice * a.kchunk, kend = min(a.K, kbeg + a.kchunk);
  const int T = (kend - kbeg + WG_TK - 1) / WG_TK;             // >= 1: no slice is empty (plan_wgrad_slices)
  const bool vec_a = (P.lda & 3) == 0 && (((size_t)P.A) & 15) == 0;
  const bool vec_d = (P.ldd & 3) == 0 && (((size_t)P.D) & 15) == 0;
  const bool ones = tm == 0;                                   // the m-tile-0 workgroups also sum the columns of B

  f32x16 acc[2], acc2[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; acc2[0][r] = 0.f; acc2[1][r] = 0.f; }
  float cs = 0.f, cs2 = 0.f;

  WgradRegs regs;
  wgrad_load(P, a.w, m0, n0, kbeg, kend, tid, vec_a, vec_d, regs);
  wgrad_stage(smem, tid, regs);
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const float* st = smem + (t & 1) * WG_STAGE;
    const float* As = st;
    const float* Bs = st + WG_TK * WG_LDA;
    const float* Bs2 = Bs + WG_TK * WG_LDB;
    // tile t + 1 travels to registers under the MFMAs of tile t (clamped: the last iteration re-reads its own)
    wgrad_load(P, a.w, m0, n0, kbeg + min(t + 1, T - 1) * WG_TK, kend, tid, vec_a, vec_d, regs);
    if (ones) {       // wave w adds the k rows 8w .. 8w+7 of the columns it sees: 16 LDS reads per k-tile
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        cs += Bs[(8 * wave + kk) * WG_LDB + lane];
        cs2 += Bs2[(8 * wave + kk) * WG_LDB + lane];
      }
    }
#pragma unroll
    for (int kk = 0; kk < WG_TK; kk += 2) {
      const int kr = kk + (lane >> 5);
      const float a0 = As[kr * WG_LDA + wm * 64 + (lane & 31)];
      const float a1 = As[kr * WG_LDA + wm * 64 + 32 + (lane & 31)];
      const float b = Bs[kr * WG_LDB + wn * 32 + (lane & 31)];
      const float b2 = Bs2[kr * WG_LDB + wn * 32 + (lane & 31)];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1], 0, 0, 0);
      acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b2, acc2[0], 0, 0, 0);
      acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b2, acc2[1], 0, 0, 0);
    }
    if (t + 1 < T) wgrad_stage(smem + ((t + 1) & 1) * WG_STAGE, tid, regs);
    __syncthreads();
  }
  // column sums of the four waves, fixed order
  float* s_cs = smem + 2 * WG_STAGE;          // [2][4][64]
  if (ones) {
    s_cs[wave * 64 + lane] = cs;
    s_cs[256 + wave * 64 + lane] = cs2;
  }
  __syncthreads();
  if (ones && tid < 128) {
    const float* q = s_cs + (tid >> 6) * 256 + (tid & 63);
    cs = (q[0] + q[64]) + (q[128] + q[192]);    // tid < 64: column sums of B; 64 .. 127: of w (.) B
  }

  const long long ldc = P.n_out;
  float* c1 = a.g1 + P.c_off;
  float* c2 = a.g2 + P.c_off;
  if (a.slices == 1) {                          // one slice: straight into the accumulators
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      int m, n;
      wgrad_mn(tid, i, m, n);
      m += m0; n += n0;
      if (m < P.k_in && n < P.n_out) {
        const float v1 = acc[i >> 4][i & 15], v2 = acc2[i >> 4][i & 15];
        float* q1 = c1 + (long long)m * ldc + n;
        float* q2 = c2 + (long long)m * ldc + n;
        if (a.fresh) { *q1 = v1; *q2 = v2; } else { *q1 += v1; *q2 += v2; }
      }
    }
    if (ones && tid < 128 && n0 + (tid & 63) < P.n_out) {
      float* q = (tid < 64 ? c1 : c2) + (long long)P.k_in * ldc + n0 + (tid & 63);
      if (a.fresh) *q = cs; else *q += cs;
    }
    return;
  }
  // partial tile -> workspace, in accumulator order
  float* part = a.ws + ((long long)tile * a.slices + slice) * 2 * WG_PART;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    part[i * 256 + tid] = acc[i >> 4][i & 15];
    part[WG_PART + i * 256 + tid] = acc2[i >> 4][i & 15];
  }
  if (ones && tid < 128) part[(tid >> 6) * WG_PART + WG_TM * WG_TN + (tid & 63)] = cs;
  // ticket: the partial is released device-wide before the counter moves; whoever draws the last ticket
  // acquires everybody's
  __threadfence();
  __syncthreads();
  int* s_ticket = (int*)(s_cs + 512);
  if (tid == 0) *s_ticket = __hip_atomic_fetch_add(a.tickets + tile, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (*s_ticket != a.slices - 1) return;        // block-uniform
  __threadfence();
  const float* base = a.ws + (long long)tile * a.slices * 2 * WG_PART;
#pragma unroll 4
  for (int i = 0; i < 32; ++i) {
    int m, n;
    wgrad_mn(tid, i, m, n);
    m += m0; n += n0;
    float v1 = 0.f, v2 = 0.f;
    for (int z = 0; z < a.slices; ++z) {        // fixed order, whoever folds
      const float* q = base + (long long)z * 2 * WG_PART + i * 256 + tid;
      v1 += __builtin_nontemporal_load(q);
      v2 += __builtin_nontemporal_load(q + WG_PART);
    }
    if (m < P.k_in && n < P.n_out) {
      float* q1 = c1 + (long long)m * ldc + n;
      float* q2 = c2 + (long long)m * ldc + n;
      if (a.fresh) { *q1 = v1; *q2 = v2; } else { *q1 += v1; *q2 += v2; }
    }
  }
  if (ones && tid < 128) {
    float v = 0.f;
    for (int z = 0; z < a.slices; ++z)
      v += __builtin_nontemporal_load(base + ((long long)z * 2 + (tid >> 6)) * WG_PART + WG_TM * WG_TN + (tid & 63));
    if (n0 + (tid & 63) < P.n_out) {
      float* q = (tid < 64 ? c1 : c2) + (long long)P.k_in * ldc + n0 + (tid & 63);
      if (a.fresh) *q = v; else *q += v;
    }
  }
  if (tid == 0) a.tickets[tile] = 0;            // for the next launch (stream-ordered after this one)
}

// N = 1 problems: C[m] = sum_b A(b, m) s_b and sum_b A(b, m) w_b s_b for 64 columns m (m == k_in: the ones
// row), s_b = D[b * ldd]; 4 row groups x 64 columns, rows b = rg, rg + 4, ...; groups added as (0+1)+(2+3)
__device__ __forceinline__ void wgrad_colsum(const WgradArgs& a, int cblock, float* smem) {
  const int tid = threadIdx.x, col = tid & 63, rg = tid >> 6;
  int pi = 0;
  for (int i = 0; i < a.n_prob; ++i)
    if (a.prob[i].n_out <= 1 && cblock >= a.prob[i].col0) pi = i;      // block-uniform
  const WgradProblem P = a.prob[pi];
  const int m = (cblock - P.col0) * 64 + col;
  const bool real = m < P.k_in, one = m == P.k_in;
  const float* ap = P.A + (real ? m : 0);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
  for (int b = rg; b < a.K; b += 4) {
    const float av = real ? ap[(long long)b * P.lda] : (one ? 1.f : 0.f);
    const float sv = P.D[(long long)b * P.ldd];
    s1 += av * sv;
    s2 += av * (a.w[b] * sv);
  }
  smem[rg * 64 + col] = s1;
  smem[256 + rg * 64 + col] = s2;
  __syncthreads();
  if (tid < 128 && (real || one)) {
    const float* q = smem + (tid >> 6) * 256 + col;
    const float v = (q[0] + q[64]) + (q[128] + q[192]);
    float* dst = (tid < 64 ? a.g1 : a.g2) + P.c_off + m;
    if (a.fresh) *dst = v; else *dst += v;
  }
}

__global__ __launch_bounds__(256) void k_wgrad(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wg_smem[];
  const int b = blockIdx.x;
  if (b < a.mfma_blocks) {
    const WgradBlock m = plan_wgrad_block(b, a.tiles, a.slices);
    if (m.slice < 0) return;
    wgrad_tile(a, m.slice, m.tile, wg_smem);
  } else if (b < a.mfma_blocks + a.col_blocks) {
    wgrad_colsum(a, b - a.mfma_blocks, wg_smem);
  } else if (a.job.sc) {
    scalar_accum_body(a.job, a.fresh != 0, (double*)wg_smem, (double*)wg_smem + 256);
  }
}

size_t wgrad_lds_bytes() { return sizeof(float) * (2 * WG_STAGE + 512 + 4); }

hipError_t launch_wgrad(hipStream_t s, const WgradLaunch& L) {
  WgradArgs a;
  memset((void*)&a, 0, sizeof(a));
  a.prob = (const WgradProblem*)L.dev_problems; a.n_prob = L.n_prob;
  a.tiles = L.tiles; a.slices = L.slices; a.kchunk = plan_wgrad_kchunk(L.K, L.slices);
  a.mfma_blocks = plan_wgrad_grid(L.tiles, L.slices); a.col_blocks = L.col_blocks;
  a.K = L.K; a.w = L.w; a.g1 = L.g1; a.g2 = L.g2; a.ws = L.ws; a.tickets = L.tickets; a.fresh = L.fresh ? 1 : 0;
  a.job = ScalarJob{L.sc_eloc, L.sc_ratio, L.sc_out, L.sc_B, L.sc_mode};
  const int grid = a.mfma_blocks + a.col_blocks + (L.sc_out ? 1 : 0);
  if (grid <= 0 || L.K <= 0) return hipSuccess;
  static bool attr_set = false;
  const size_t lds = wgrad_lds_bytes();
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_wgrad, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(k_wgrad, dim3(grid), dim3(256), lds, s, a);
  return hipGetLastError();
}

size_t wgrad_problem_bytes() { return sizeof(WgradProblem); }

void wgrad_fill_problem(void* dst, int index, const float* A, long long lda, const float* D, long long ldd,
                        long long c_off, int k_in, int n_out, int tile0, int col0) {
  WgradProblem& p = ((WgradProblem*)dst)[index];
  p.A = A; p.lda = lda; p.D = D; p.ldd = ldd; p.c_off = c_off; p.k_in = k_in; p.n_out = n_out;
  p.tile0 = tile0; p.tiles_n = (n_out + WG_TN - 1) / WG_TN; p.col0 = col0;
}

__global__ __launch_bounds__(256) void k_gemm_reduce(GemmArgs g) {
  const long long mn = (long long)g.M * g.N;
  const int nd = g.dual ? 2 : 1;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < mn * nd;
       i += (long long)gridDim.x * 256) {
    const int d = (int)(i / mn);
    const long long e = i % mn;
    float v = 0.f;
    for (int z = 0; z < g.splitk; ++z) v += g.workspace[((long long)z * nd + d) * mn + e];
    gemm_epilogue(g, d ? g.C2 : g.C, (int)(e / g.N), (int)(e % g.N), v);
  }
}

hipError_t launch_gemm(hipStream_t s, const GemmArgs& g) {
  if (g.M <= 0 || g.N <= 0) return hipSuccess;
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
