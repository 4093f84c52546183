// Stochastic-reconfiguration matrix-vector product on large-tile fp32-MFMA GEMMs (extension named
// by the north star; no reference counterpart -- see sr.hip).
//
// One CG iteration needs, over all R stored samples, t_b = O_b . p and u = sum_b t_b O_b with
// O_b = d logit_b / d theta.  For a dense layer O_b restricted to (W_l, b_l) is
// (a_{l-1}[b] (x) delta_l[b], delta_l[b]) with the stored activations a and back-propagated
// delta_l = d logit / d z_l, so
//   t_b  = sum_l delta_l[b] . (a_{l-1}[b] V_l + v_l)          (V_l, v_l = the (W_l, b_l) slice of p)
//   u_Wl = a_{l-1}^T (t (.) delta_l),   u_bl = sum_b t_b delta_l[b]
// i.e. one [R, K] x [K, H] product with a row-dot epilogue and one [K, R] x [R, H] product per
// layer -- 2 F_amp flops per sample (the forward-mode tangent chain this replaces needs 3), and no
// activation derivative at all (delta already carries it).
//
//   k_sr_rowdot  128 samples x 256 units per workgroup (8 waves as 4 x 2, 1 x 4 accumulators of
//                v_mfma_f32_32x32x2_f32 each), K in steps of 16 through double-buffered LDS tiles;
//                the weight slice (<= 256 KB) is re-read from L2 by every workgroup, the
//                activations stream from HBM once.  Epilogue: t[row] (+)= sum_n (C + v)[row][n]
//                delta[row][n], combined in a fixed order (no atomics).
//   k_sr_wsum    256 x 256 output tile per workgroup (8 waves as 2 x 4, 4 x 2 accumulators), the
//                reduction runs over a slice of the samples (steps of 8); partial products go to a
//                workspace that k_sr_wsum_reduce folds in slice order.
#include "common.hpp"
#include <cstring>
#include <cstdlib>

// several independent row-dot problems (the layers / column blocks of one matvec, each writing its
// own t) in ONE launch: a single 204,800-sample problem is 1600 tiles = 6.25 rounds of 256 CUs, i.e.
// a seventh round with a quarter of the chip; three of them back to back are 18.75 rounds.
// Workgroup id -> (problem, tile) through first_tile (dispatch order: largest K first).
// (RD_MAXB, problems per launch: plan.hpp)
struct SrRowdotBatch {
  SrRowdotArgs g[RD_MAXB];
  int first_tile[RD_MAXB + 1];
  int count;
};

struct SrWsumArgs {
  const float* A; long long lda;      // A(k, m) = A[k * lda + m]
  const float* D; long long ldd;      // delta(k, n)
  const float* t;                     // [R] row scale of delta
  float* ws;                          // [slices][(M + 1) * N]
  float* out; long long ldo;          // out(m, n) = out[m * ldo + n]: a block of the W rows (theta layout)
  float* bias_out;                    // [N] bias row of these columns, or nullptr (row blocks m0 > 0)
  int M, N, R, slices;
};

namespace {

// (RD_TM = 128 samples per tile: plan.hpp)
#define RD_TN 256
#define RD_K 16
#define RD_LDA (RD_TM + 4)
#define RD_LDB (RD_TN + 4)

// VEC: every row stride / extent is a multiple of 4 floats and the bases are 16-byte aligned: all
// global loads are unconditional float4 loads from clamped (always valid) addresses whose
// out-of-range results are replaced by zeros -- straight-line code, so the compiler counts vmcnt
// exactly and the tile requested two K-steps ahead really stays in flight under the MFMAs (a load
// under a run-time branch makes every later use wait for ALL outstanding loads).  !VEC is the
// general element-wise path.
template <bool VEC>
__global__ __launch_bounds__(512, 2) void k_sr_rowdot(SrRowdotBatch bt) {
  __shared__ __attribute__((aligned(16))) float As[2][RD_K][RD_LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][RD_K][RD_LDB];
  __shared__ float s_part[2][RD_TM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int slot = 0;
  while (slot + 1 < bt.count && (int)blockIdx.x >= bt.first_tile[slot + 1]) ++slot;   // block-uniform
  const SrRowdotArgs g = bt.g[slot];
  const int m0 = ((int)blockIdx.x - bt.first_tile[slot]) * RD_TM;
  const int T = (g.K + RD_K - 1) / RD_K;

  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // operand tiles in registers: A 128 x 16 (one float4 along k per thread), V 16 x 256 (two);
  // two register sets: tile t+2 is requested while tile t is multiplied
  struct Regs { f32x4 ra, rv[2]; };
  Regs set0, set1;
  const int am = tid >> 2, ak = (tid & 3) * 4;
  auto request = [&](int t, Regs& q) {
    const int k0 = min(t, T - 1) * RD_K;
    if (VEC) {
      const int m = m0 + am, k = k0 + ak;
      // out-of-range elements are zeroed by a multiply, not a select: the compiler turns a select
      // into a branch around the load, which puts a full vmcnt(0) wait back into the loop
      const f32x4 v = *(const f32x4*)(g.A + (long long)min(m, g.M - 1) * g.lda + min(k, g.K - 4));
      q.ra = v * ((m < g.M && k < g.K) ? 1.f : 0.f);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = tid + 512 * i, kv = k0 + (idx >> 6), n = (idx & 63) * 4;
        const f32x4 w = *(const f32x4*)(g.V + (long long)min(kv, g.K - 1) * g.ldv + min(n, g.N - 4));
        q.rv[i] = w * ((kv < g.K && n < g.N) ? 1.f : 0.f);
      }
    } else {
      const int m = m0 + am, k = k0 + ak;
      const float* p = g.A + (long long)m * g.lda + k;
#pragma unroll
      for (int j = 0; j < 4; ++j) q.ra[j] = (m < g.M && k + j < g.K) ? p[j] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = tid + 512 * i, kv = k0 + (idx >> 6), n = (idx & 63) * 4;
        const float* pv = g.V + (long long)kv * g.ldv + n;
#pragma unroll
        for (int j = 0; j < 4; ++j) q.rv[i][j] = (kv < g.K && n + j < g.N) ? pv[j] : 0.f;
      }
    }
  };
  auto stage = [&](int buf, const Regs& q) {
#pragma unroll
    for (int j = 0; j < 4; ++j) As[buf][ak + j][am] = q.ra[j];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 512 * i;
      *(f32x4*)&Bs[buf][idx >> 6][(idx & 63) * 4] = q.rv[i];
    }
  };
  // tile t: multiplied from LDS buffer t & 1; tile t+1 goes from register set `nxt` to the other
  // buffer (last read before the previous barrier); tile t+2 is requested into `cur`'s set.
  // The fragment reads of k-pair kk+2 are issued before the MFMAs of k-pair kk.
  auto iter = [&](int t, Regs& cur, Regs& nxt) {
    const int buf = t & 1;
    request(t + 2, cur);                 // unconditional (clamped): exact vmcnt
    // MFMA column j of accumulator nt stands for memory column wn*128 + 4 j + nt: the four B
    // fragments of a lane are ONE ds_read_b128 (conflict-free), and the epilogue's delta / bias
    // operands of a row are one float4 per lane
    float av[2]; f32x4 bv[2];
    auto frags = [&](int kk, int st) {
      av[st] = As[buf][kk + (lane >> 5)][wm * 32 + (lane & 31)];
      bv[st] = *(const f32x4*)&Bs[buf][kk + (lane >> 5)][wn * 128 + 4 * (lane & 31)];
    };
    frags(0, 0);
#pragma unroll
    for (int kk = 0; kk < RD_K; kk += 2) {
      const int st = (kk >> 1) & 1;
      if (kk + 2 < RD_K) frags(kk + 2, st ^ 1);
      __builtin_amdgcn_sched_barrier(0);   // the next k-pair's reads stay ahead of these MFMAs
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st], bv[st][nt], acc[nt], 0, 0, 0);
    }
    if (t + 1 < T) stage(buf ^ 1, nxt);
    __syncthreads();
  };
  request(0, set0);
  stage(0, set0);
  request(1, set1);
  __syncthreads();
  for (int t = 0; t < T; t += 2) {
    iter(t, set0, set1);
    if (t + 1 < T) iter(t + 1, set1, set0);
  }
  // row-dot epilogue: lane holds columns n0 + nt (nt = 0..3), n0 = wn*128 + 4 (lane & 31), of rows
  // m = wm*32 + (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  float part[16];
  {
    const int n0 = wn * 128 + 4 * (lane & 31);
    f32x4 bias, d[16];
    if (VEC) {
      const float live = n0 < g.N ? 1.f : 0.f;
      bias = *(const f32x4*)(g.vb + min(n0, g.N - 4));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        d[r] = *(const f32x4*)(g.D + (long long)min(m, g.M - 1) * g.ldd + min(n0, g.N - 4)) *
               (m < g.M ? live : 0.f);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) bias[e] = n0 + e < g.N ? g.vb[n0 + e] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
        for (int e = 0; e < 4; ++e) d[r][e] = (m < g.M && n0 + e < g.N) ? g.D[(long long)m * g.ldd + n0 + e] : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = (acc[0][r] + bias[0]) * d[r][0];
#pragma unroll
      for (int nt = 1; nt < 4; ++nt) v = fmaf(acc[nt][r] + bias[nt], d[r][nt], v);
      part[r] = v;
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) part[r] += __shfl_xor(part[r], d);
    if ((lane & 31) == 0) s_part[wn][wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = part[r];
  }
  __syncthreads();
  if (tid < RD_TM && m0 + tid < g.M) {
    const float v = s_part[0][tid] + s_part[1][tid];
    if (g.first) g.t[m0 + tid] = v; else g.t[m0 + tid] += v;
  }
}

#define WS_T 256
#define WS_K 8
#define WS_LD (WS_T + 4)

template <bool VEC>
__global__ __launch_bounds__(512, 2) void k_sr_wsum(SrWsumArgs g) {
  __shared__ __attribute__((aligned(16))) float As[2][WS_K][WS_LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][WS_K][WS_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;          // 2 x 4 waves: 128 rows x 64 columns each
  int per = (g.R + g.slices - 1) / g.slices;
  per = (per + WS_K - 1) / WS_K * WS_K;
  const int kbeg = blockIdx.x * per, kend = min(g.R, kbeg + per);
  const int T = kbeg < kend ? (kend - kbeg + WS_K - 1) / WS_K : 0;
  const bool m_live = wm * 128 < g.M, n_live = wn * 64 < g.N;     // wave-uniform

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // bias row: thread (lk, lc) sums t_b delta[b][lc..lc+3] over its samples b = lk (mod 8); the
  // eight row groups are combined through LDS at the end (fixed order)
  f32x4 cs = {0.f, 0.f, 0.f, 0.f};

  struct Regs { f32x4 ra, rd; };
  Regs set0, set1;
  const int lk = tid >> 6, lc = (tid & 63) * 4;      // row (sample) of the K-step, first column
  auto request = [&](int t, Regs& q) {
    const int k = kbeg + min(t, max(T - 1, 0)) * WS_K + lk;
    if (VEC) {
      const int kc = min(k, g.R - 1);
      const f32x4 a = *(const f32x4*)(g.A + (long long)kc * g.lda + min(lc, g.M - 4));
      const f32x4 d = *(const f32x4*)(g.D + (long long)kc * g.ldd + min(lc, g.N - 4));
      const float sc = g.t[kc];
      q.ra = a * ((k < kend && lc < g.M) ? 1.f : 0.f);
      q.rd = d * ((k < kend && lc < g.N) ? sc : 0.f);
    } else {
      const float* pa = g.A + (long long)k * g.lda + lc;
      const float* pd = g.D + (long long)k * g.ldd + lc;
      const float sc = k < kend ? g.t[k] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        q.ra[j] = (k < kend && lc + j < g.M) ? pa[j] : 0.f;
        q.rd[j] = (k < kend && lc + j < g.N) ? pd[j] * sc : 0.f;
      }
    }
  };
  auto stage = [&](int buf, const Regs& q) {
    *(f32x4*)&As[buf][lk][lc] = q.ra;
    *(f32x4*)&Bs[buf][lk][lc] = q.rd;
    cs += q.rd;
  };
  auto iter = [&](int t, Regs& cur, Regs& nxt) {
    const int buf = t & 1;
    request(t + 2, cur);
    if (m_live && n_live) {
      float av[2][4], bv[2][2];
      auto frags = [&](int kk, int st) {
#pragma unroll
        for (int i = 0; i < 4; ++i) av[st][i] = As[buf][kk + (lane >> 5)][wm * 128 + i * 32 + (lane & 31)];
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[st][j] = Bs[buf][kk + (lane >> 5)][wn * 64 + j * 32 + (lane & 31)];
      };
      frags(0, 0);
#pragma unroll
      for (int kk = 0; kk < WS_K; kk += 2) {
        const int st = (kk >> 1) & 1;
        if (kk + 2 < WS_K) frags(kk + 2, st ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st][i], bv[st][j], acc[i][j], 0, 0, 0);
      }
    }
    if (t + 1 < T) stage(buf ^ 1, nxt);
    __syncthreads();
  };
  if (T > 0) {
    request(0, set0);
    stage(0, set0);
    request(1, set1);
  }
  __syncthreads();
  for (int t = 0; t < T; t += 2) {
    iter(t, set0, set1);
    if (t + 1 < T) iter(t + 1, set1, set0);
  }
  float* ws = g.ws + (long long)blockIdx.x * (g.M + 1) * g.N;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = wn * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < g.M && n < g.N) ws[(long long)m * g.N + n] = acc[i][j][r];
      }
    }
  // the bias row: sum_b t_b delta[b]
  *(f32x4*)&As[0][lk][lc] = cs;
  __syncthreads();
  if (tid < g.N) {
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < WS_K; ++r) v += As[0][r][tid];
    ws[(long long)g.M * g.N + tid] = v;
  }
}

// out[i] = sum over the slices of ws[s][i], slices in a fixed order; one thread per output so that
// every load is coalesced, eight independent partial sums in flight
__global__ __launch_bounds__(256) void k_sr_wsum_reduce(SrWsumArgs g) {
  const long long mn = (long long)(g.M + 1) * g.N;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < mn; i += (long long)gridDim.x * 256) {
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 8 <= g.slices; s += 8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += g.ws[(long long)(s + e) * mn + i];
    }
    for (; s < g.slices; ++s) v[0] += g.ws[(long long)s * mn + i];
    const float sum = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    const long long m = i / g.N, n = i % g.N;
    if (m < g.M) g.out[m * g.ldo + n] = sum;
    else if (g.bias_out) g.bias_out[n] = sum;
  }
}

// t_b += a_L[b] . v_out + v_bout (fully_connected output layer) or x_b . v_on + v_bon (rbm onsite
// layer): one wave per sample, four samples (16 loads) in flight per wave, grid-stride.  Per sample
// the arithmetic is fixed: lane l sums k = l, l + 64, ... in order, then the xor tree.
__global__ __launch_bounds__(256) void k_sr_row_linear(const float* __restrict__ x, long long ldx,
                                                       const float* __restrict__ v,
                                                       const float* __restrict__ vb, int R, int K,
                                                       float* __restrict__ t,
                                                       const float* __restrict__ parts, int nparts,
                                                       long long pstride) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const float bias = vb[0];
  for (int b0 = 4 * wave; b0 < R; b0 += 4 * nwaves) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = lane; k < K; k += 64) {
      const float vk = v[k];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int b = min(b0 + j, R - 1);              // rows past the end repeat the last one (not stored)
        s[j] = fmaf(x[(long long)b * ldx + k], vk, s[j]);
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] += __shfl_xor(s[j], d);
    }
    if (lane < 4 && b0 + lane < R) {
      // nparts > 0: t = ((part_0 + part_1) + ...) + this term -- the order in which the row-dot
      // launches used to add into t one after the other
      const int b = b0 + lane;
      float acc = nparts > 0 ? parts[b] : t[b];
      for (int j = 1; j < nparts; ++j) acc += parts[(long long)j * pstride + b];
      t[b] = acc + ((lane == 0 ? s[0] : lane == 1 ? s[1] : lane == 2 ? s[2] : s[3]) + bias);
    }
  }
}

// u[0..K) = sum_b t_b x[b][0..K), u[K] = sum_b t_b (the N = 1 layer: [x | 1]^T t), and
// tsum = sum_b t_b.  Two stages, fixed order: slices of the samples, then the slices.  A lane
// carries four neighbouring columns (one 16-byte load per row: a wave reads a whole 256-column row
// at once, K <= 256 here); wave w takes the samples b0 + w, b0 + w + 4, ..., four rows in flight,
// each with its own accumulator -- the per-column order of additions is what it always was.
#define CS_THREADS 256
template <bool VEC>
__global__ __launch_bounds__(CS_THREADS) void k_sr_colsum(const float* __restrict__ x, long long ldx,
                                                          const float* __restrict__ t, int R, int K,
                                                          float* __restrict__ ws /*[slices][K + 1]*/) {
  __shared__ f32x4 s_acc[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per = (R + gridDim.x - 1) / gridDim.x;
  const int b0 = blockIdx.x * per, b1 = min(R, b0 + per);
  float* out = ws + (long long)blockIdx.x * (K + 1);
  if (VEC) {
    // (row stride a multiple of 4 floats, base 16-byte aligned: a vector that starts inside a row
    // stays inside it; lanes past the row repeat its last vector and store nothing)
    for (int k0 = 0; k0 < K; k0 += 256) {
      const int kc = min(k0 + 4 * lane, (int)ldx - 4);
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      f32x4 a0 = z, a1 = z, a2 = z, a3 = z;
      int b = b0 + wave;
      for (; b + 12 < b1; b += 16) {
        const f32x4 x0 = *(const f32x4*)(x + (long long)b * ldx + kc), x1 = *(const f32x4*)(x + (long long)(b + 4) * ldx + kc);
        const f32x4 x2 = *(const f32x4*)(x + (long long)(b + 8) * ldx + kc), x3 = *(const f32x4*)(x + (long long)(b + 12) * ldx + kc);
        const float t0 = t[b], t1 = t[b + 4], t2 = t[b + 8], t3 = t[b + 12];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a0[e] = fmaf(t0, x0[e], a0[e]); a1[e] = fmaf(t1, x1[e], a1[e]);
          a2[e] = fmaf(t2, x2[e], a2[e]); a3[e] = fmaf(t3, x3[e], a3[e]);
        }
      }
      for (; b < b1; b += 4) {
        const f32x4 x0 = *(const f32x4*)(x + (long long)b * ldx + kc);
        const float t0 = t[b];
#pragma unroll
        for (int e = 0; e < 4; ++e) a0[e] = fmaf(t0, x0[e], a0[e]);
      }
      s_acc[wave][lane] = (a0 + a1) + (a2 + a3);
      __syncthreads();
      if (wave == 0 && k0 + 4 * lane == kc) {
        const f32x4 r = (s_acc[0][lane] + s_acc[1][lane]) + (s_acc[2][lane] + s_acc[3][lane]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (kc + e < K) out[kc + e] = r[e];
      }
      __syncthreads();
    }
  } else {
    // any row stride: 64 columns at a time, one float per lane
    float* s_a = (float*)s_acc;
    for (int k0 = 0; k0 < K; k0 += 64) {
      const int k = min(k0 + lane, K - 1);
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int b = b0 + wave;
      for (; b + 12 < b1; b += 16) {
        const float x0 = x[(long long)b * ldx + k], x1 = x[(long long)(b + 4) * ldx + k];
        const float x2 = x[(long long)(b + 8) * ldx + k], x3 = x[(long long)(b + 12) * ldx + k];
        a0 = fmaf(t[b], x0, a0); a1 = fmaf(t[b + 4], x1, a1);
        a2 = fmaf(t[b + 8], x2, a2); a3 = fmaf(t[b + 12], x3, a3);
      }
      for (; b < b1; b += 4) a0 = fmaf(t[b], x[(long long)b * ldx + k], a0);
      s_a[wave * 64 + lane] = (a0 + a1) + (a2 + a3);
      __syncthreads();
      if (wave == 0 && k0 + lane < K)
        out[k0 + lane] = (s_a[lane] + s_a[64 + lane]) + (s_a[128 + lane] + s_a[192 + lane]);
      __syncthreads();
    }
  }
  __shared__ float s_t[CS_THREADS];
  float a = 0.f;
  for (int b = b0 + threadIdx.x; b < b1; b += CS_THREADS) a += t[b];
  s_t[threadIdx.x] = a;
  __syncthreads();
  for (int d = CS_THREADS / 2; d >= 1; d >>= 1) {
    if ((int)threadIdx.x < d) s_t[threadIdx.x] += s_t[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[K] = s_t[0];
}

// one wave per output: lanes over the slices
__global__ __launch_bounds__(256) void k_sr_colsum_reduce(const float* __restrict__ ws, int slices, int K,
                                                          float* __restrict__ u, float* __restrict__ tsum) {
  const int lane = threadIdx.x & 63;
  for (int k = blockIdx.x * 4 + (threadIdx.x >> 6); k <= K; k += gridDim.x * 4) {
    float v = 0.f;
    for (int s = lane; s < slices; s += 64) v += ws[(long long)s * (K + 1) + k];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if (lane == 0) { u[k] = v; if (k == K) tsum[0] = v; }
  }
}

}  // namespace

hipError_t launch_sr_colsum(hipStream_t s, const float* x, long long ldx, const float* t, int R,
                            int K, float* ws, int slices, float* u, float* tsum) {
  if (ldx % 4 == 0 && ldx >= 4 && ((size_t)x & 15) == 0)
    hipLaunchKernelGGL(k_sr_colsum<true>, dim3(slices), dim3(CS_THREADS), 0, s, x, ldx, t, R, K, ws);
  else
    hipLaunchKernelGGL(k_sr_colsum<false>, dim3(slices), dim3(CS_THREADS), 0, s, x, ldx, t, R, K, ws);
  hipLaunchKernelGGL(k_sr_colsum_reduce, dim3((K + 4) / 4), dim3(256), 0, s, ws, slices, K, u, tsum);
  return hipGetLastError();
}

static bool rowdot_vec(const SrRowdotArgs& g) {
  return (g.lda & 3) == 0 && (g.ldv & 3) == 0 && (g.K & 3) == 0 && (g.N & 3) == 0 && g.K >= 4 && g.N >= 4 &&
         (((size_t)g.A | (size_t)g.V) & 15) == 0;
}

// count problems in as few launches as possible: one when they agree on the vector path and count
// <= RD_MAXB (the usual case), else one per run of compatible problems
hipError_t launch_sr_rowdot_batch(hipStream_t s, const SrRowdotArgs* probs, int count) {
  int i = 0;
  while (i < count) {
    const bool vec = rowdot_vec(probs[i]);
    int n = 1;
    while (i + n < count && n < RD_MAXB && rowdot_vec(probs[i + n]) == vec) ++n;
    // dispatch order: largest K first (list scheduling: the light tiles fill the end) -- plan.hpp
    int order[RD_MAXB], ks[RD_MAXB], ms[RD_MAXB];
    for (int j = 0; j < n; ++j) { ks[j] = probs[i + j].K; ms[j] = probs[i + j].M; }
    SrRowdotBatch bt;
    memset((void*)&bt, 0, sizeof(bt));
    bt.count = n;
    const int tiles = plan_rowdot_schedule(ks, ms, n, order, bt.first_tile);
    for (int j = 0; j < n; ++j) {
      const SrRowdotArgs& g = probs[i + order[j]];
      if (g.N > RD_TN) return hipErrorInvalidValue;
      bt.g[j] = g;
    }
    if (tiles > 0) {
      if (vec) hipLaunchKernelGGL(k_sr_rowdot<true>, dim3(tiles), dim3(512), 0, s, bt);
      else hipLaunchKernelGGL(k_sr_rowdot<false>, dim3(tiles), dim3(512), 0, s, bt);
    }
    i += n;
  }
  return hipGetLastError();
}

hipError_t launch_sr_rowdot(hipStream_t s, const float* A, long long lda, const float* V,
                            long long ldv, const float* vb, const float* D, long long ldd, float* t,
                            int M, int N, int K, bool first) {
  SrRowdotArgs g{A, lda, V, ldv, vb, D, ldd, t, M, N, K, first ? 1 : 0};
  return launch_sr_rowdot_batch(s, &g, 1);
}

int sr_wsum_slices(int R, int num_cus) { return plan_sr_wsum_slices(R, num_cus); }

// one (<= 256 x <= 256) block of u_W = A^T (t (.) D); wider layers are tiled by the caller
hipError_t launch_sr_wsum(hipStream_t s, const float* A, long long lda, const float* D,
                          long long ldd, const float* t, float* ws, float* out, long long ldo,
                          float* bias_out, int M, int N, int R, int slices) {
  if (M > WS_T || N > WS_T) return hipErrorInvalidValue;
  SrWsumArgs g{A, lda, D, ldd, t, ws, out, ldo, bias_out, M, N, R, slices};
  const bool vec = (lda & 3) == 0 && (ldd & 3) == 0 && (M & 3) == 0 && (N & 3) == 0 && M >= 4 && N >= 4 &&
                   (((size_t)A | (size_t)D) & 15) == 0;
  if (vec) hipLaunchKernelGGL(k_sr_wsum<true>, dim3(slices), dim3(512), 0, s, g);
  else hipLaunchKernelGGL(k_sr_wsum<false>, dim3(slices), dim3(512), 0, s, g);
  const long long mn = (long long)(M + 1) * N;
  const int blocks = (int)((mn + 255) / 256);
  hipLaunchKernelGGL(k_sr_wsum_reduce, dim3(blocks), dim3(256), 0, s, g);
  return hipGetLastError();
}

hipError_t launch_sr_row_linear(hipStream_t s, const float* x, long long ldx, const float* v,
                                const float* vb, int R, int K, float* t, const float* parts, int nparts,
                                long long pstride) {
  const int blocks = (R + 15) / 16 < 2048 ? (R + 15) / 16 : 2048;      // 4 waves x 4 samples per pass
  hipLaunchKernelGGL(k_sr_row_linear, dim3(blocks), dim3(256), 0, s, x, ldx, v, vb, R, K, t, parts, nparts, pstride);
  return hipGetLastError();
}
