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

// Several independent dual GEMMs (one per layer's weight gradient) in ONE launch:
// blockIdx.z = problem * splitk + split.  With ~4 workgroups co-resident per CU the global
// load latency of one is hidden behind the MFMAs of the others.
// XCD-aware block order: workgroups go to the 8 XCDs round robin by linear block id, and every XCD
// has its own L2.  All tiles of one (problem, k-slice) group -- they share that slice's A and B
// operand rows -- are given ids that are congruent mod 8, so one XCD's L2 fetches a slice once
// instead of (up to) eight L2s fetching it each: group g lives on XCD g % 8.
template <bool DUAL, int NS = 1>
__global__ __launch_bounds__(256) void k_gemm_batched(const GemmArgs* __restrict__ batch,
                                                      int splitk, int tiles_x, int tiles_y, int n_groups) {
  const int per_group = tiles_x * tiles_y;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int grp = xcd + 8 * (idx / per_group), tile = idx % per_group;
  if (grp >= n_groups) return;                                                  // block-uniform
  const int bx = tile % tiles_x, by = tile / tiles_x;
  const GemmArgs g = batch[grp / splitk];
  const int m_rows = g.ones_row ? g.M - 1 : g.M;
  if (bx * GT >= g.N || by * GT >= m_rows) return;                              // block-uniform
  gemm_block<DUAL, NS>(g, bx, by, grp % splitk);
}

// fresh: the destination holds no sum yet (accumulators after reset_gradients): epilogue 3 stores
__device__ __forceinline__ void gemm_reduce_body(const GemmArgs& g, long long start,
                                                 long long stride, bool fresh = false) {
  const long long mn = (long long)g.M * g.N;
  const int nd = g.dual ? 2 : 1;
  for (long long i = start; i < mn * nd; i += stride) {
    const int d = (int)(i / mn);
    const long long e = i % mn;
    float v = 0.f;
    for (int z = 0; z < g.splitk; ++z) v += g.workspace[((long long)z * nd + d) * mn + e];
    if (fresh && g.epilogue == 3) (d ? g.C2 : g.C)[(e / g.N) * g.ldc + (e % g.N)] = v;
    else gemm_epilogue(g, d ? g.C2 : g.C, (int)(e / g.N), (int)(e % g.N), v);
  }
}

// tf.metrics.mean updates (training.py:555, 689-690) + mean_tensor count (550-553):
// scalars = [e_total, e_count, r_total, r_count, g_count]; one workgroup, fixed order
struct ScalarJob { const float* eloc; const float* ratio; float* sc; int B, mode; };

__device__ __forceinline__ void scalar_accum_body(const ScalarJob& j, bool fresh) {
  __shared__ double se[256];
  __shared__ double sr[256];
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

// grid (blocks, count + 1): rows 0 .. count-1 fold the split-K partials of one problem each; block 0 of
// the extra row does the scalar accumulators of the same accumulate call (no launch of its own)
__global__ __launch_bounds__(256) void k_gemm_reduce_batched(const GemmArgs* __restrict__ batch, int count, int fresh,
                                                             ScalarJob job) {
  if ((int)blockIdx.y == count) {
    if (blockIdx.x == 0 && job.sc) scalar_accum_body(job, fresh != 0);
    return;
  }
  const GemmArgs g = batch[blockIdx.y];
  gemm_reduce_body(g, (long long)blockIdx.x * 256 + threadIdx.x, (long long)gridDim.x * 256, fresh != 0);
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

hipError_t launch_gemm_batched(hipStream_t s, const GemmArgs* dev_batch, int count, int max_m,
                               int max_n, int splitk, bool dual, bool fresh, const float* sc_eloc,
                               const float* sc_ratio, float* sc_out, int sc_B, int sc_mode) {
  if (count <= 0) return hipSuccess;
  const int tx = (max_n + GT - 1) / GT, ty = (max_m + GT - 1) / GT, groups = count * splitk;
  const dim3 grid(8 * ((groups + 7) / 8) * tx * ty);
  if (dual) hipLaunchKernelGGL((k_gemm_batched<true>), grid, dim3(256), 0, s, dev_batch, splitk, tx, ty, groups);
  else hipLaunchKernelGGL((k_gemm_batched<false>), grid, dim3(256), 0, s, dev_batch, splitk, tx, ty, groups);
  const long long total = (dual ? 2LL : 1LL) * (max_m + 1) * max_n;
  const int blocks = (int)min((total + 255) / 256, (long long)512);
  const ScalarJob job{sc_eloc, sc_ratio, sc_out, sc_B, sc_mode};
  hipLaunchKernelGGL(k_gemm_reduce_batched, dim3(blocks, count + (sc_out ? 1 : 0)), dim3(256), 0, s, dev_batch, count,
                     fresh ? 1 : 0, job);
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
