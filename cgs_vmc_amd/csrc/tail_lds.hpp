// Row kernel with LDS-resident operands (one barrier per layer): rows {chain, bond} of a row list ->
// logits or 0.5 jx psi'/psi (operators.py:137-169, 249-259).  Shape: 4 waves, 32 rows per tile (two
// 16-row halves that share every weight fragment).  Wave w owns output units 16 TO w .. 16 TO (w+1) - 1
// of both halves; the B operands of a layer come from LDS, the A fragments stream from L2 through a
// two-stage register ring behind a scalar base.
//   NT = 24 / 32 (384 / 512 padded hidden units): the row kernel of the fused path for
//        fc_layer_size 257 .. 512, where k_tail16's register-resident activations (2 x NT x 4
//        registers, twice) no longer fit.  Every hidden activation of layers.NONLINEARITIES (one
//        instantiation per activation object, act_tail.hip) and both dense ansatz types:
//        RBM = the last H x H layer's epilogue is sum_h log cosh(z_h) and the onsite term x . w_on
//        of the row (chain's cached value + the rank-2 exchange update) is added
//        (wavefunctions.py:418-437), as in k_tail16.
#pragma once
#include "common.hpp"

namespace tail_lds {
constexpr int NW = 4;

template <int NT, bool RATIO, bool RBM, int ACT>
__device__ __forceinline__ void tail_lds_body(const TailArgs& a) {
  constexpr int Hp = NT * 16, TO = NT / NW;
  constexpr int XBUF = 2 * NT * 256;   // floats of one operand buffer [2 halves][NT][64 lanes][4]
  static_assert(NT % NW == 0 && TO % 2 == 0, "unit tiles divide over the waves in pairs");
  extern __shared__ float smem[];
  const int n_hidden = a.n_hidden;
  float* s_x = smem;                     // [2][2][NT][64][4]
  float* s_part = s_x + 2 * XBUF;        // [NW][2][16] partial output dots
  float* s_meta = s_part + NW * 32;      // [32][3] {0.5 jx of the row's bond, logit of its chain, onsite term of the row}
  float* s_bias = s_meta + 96;           // [n_hidden][Hp]
  float* s_wout = s_bias + n_hidden * Hp;   // [Hp]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, j = lane & 15;
  const PackedParams& pp = a.pp;
  const int n_rows = a.n_rows_dev ? *a.n_rows_dev : a.n_rows;
  const int n_tiles = (n_rows + 31) >> 5;
  const float bout = pp.bout[0];
  const int oact = a.oact;
  for (int i = tid; i < n_hidden * Hp; i += 256) s_bias[i] = pp.bh[i];
  for (int i = tid; i < Hp; i += 256) s_wout[i] = pp.woutp[i];

  // weight ring: stage (ti & 1) holds k-tile ti of this wave's TO output tiles; every issue is
  // unconditional (clamped layer index) so that vmcnt can be counted exactly
  f32x4 ring[2][TO];
  // uniform (SGPR) base per output tile + one per-lane byte offset register
  typedef const __attribute__((address_space(1))) char* gchar_p;
  typedef const __attribute__((address_space(1))) f32x4* gf32x4_p;
  const unsigned lane_off = (unsigned)lane * 16u;
  const char* p16w = (const char*)pp.p16 + (size_t)wave * TO * NT * 256 * sizeof(float);
  auto issue = [&](int l, int ti, int st) {
    const char* lb = p16w + (size_t)l * Hp * Hp * sizeof(float);
    asm volatile("" : "+s"(lb));   // keep the layer base scalar (the allocator otherwise widens it per lane)
#pragma unroll
    for (int to = 0; to < TO; ++to) {
      gchar_p base = (gchar_p)lb + (size_t)(to * NT + ti) * 256 * sizeof(float);
      ring[st][to] = *(gf32x4_p)(base + lane_off);
    }
  };
  issue(0, 0, 0);

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    // ---- first-layer activations of the 32 rows (this wave's units): relu(z1[chain] -+ 2 (W1[i] - W1[j]))
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int row = tile * 32 + 16 * hf + j;
      const int2 ri = a.rowinfo[row < n_rows ? row : n_rows - 1];   // {chain, +-(bond+1) or 0}
      const int bs = ri.y;
      const int bond = (bs > 0 ? bs : -bs) - (bs != 0 ? 1 : 0);
      const float coef = bs > 0 ? -2.f : (bs < 0 ? 2.f : 0.f);      // -2 s_i, 0 for a plain row
      const int2 ab = a.bonds[bond];
      const float* zb = a.z1 + (long long)ri.x * Hp;
      const float* wa = pp.w1p + (long long)ab.x * Hp;
      const float* wb = pp.w1p + (long long)ab.y * Hp;
      if (wave == 0 && g == hf) {
        if (RATIO) {
          s_meta[(16 * hf + j) * 3] = a.half_jx[bond];
          s_meta[(16 * hf + j) * 3 + 1] = a.logit_base[ri.x];
        }
        if (RBM) s_meta[(16 * hf + j) * 3 + 2] = fmaf(coef, pp.won[ab.x] - pp.won[ab.y], a.on_base[ri.x]);
      }
#pragma unroll
      for (int t0 = 0; t0 < TO; t0 += 2) {   // two unit tiles at a time: 24 registers in flight
        f32x4 z[2], x[2], y[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int col = 16 * (wave * TO + t0 + q) + 4 * g;
          z[q] = *(const f32x4*)(zb + col);
          x[q] = *(const f32x4*)(wa + col);
          y[q] = *(const f32x4*)(wb + col);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = vmc_act<ACT>(fmaf(coef, x[q][e] - y[q][e], z[q][e]));
          *(f32x4*)(s_x + ((hf * NT + wave * TO + t0 + q) * 64 + lane) * 4) = v;
        }
        __builtin_amdgcn_sched_barrier(0);   // one batch of loads in flight at a time
      }
    }

    for (int l = 0; l < n_hidden; ++l) {
      __syncthreads();
      const f32x4* xin = (const f32x4*)(s_x + (l & 1) * XBUF) + lane;
      f32x4 acc[2][TO];
#pragma unroll
      for (int to = 0; to < TO; ++to) {
        acc[0][to] = *(const f32x4*)(s_bias + l * Hp + 16 * (wave * TO + to) + 4 * g);
        acc[1][to] = acc[0][to];
      }
      const int l_next = l + 1 < n_hidden ? l + 1 : 0;
#pragma unroll
      for (int ti = 0; ti < NT; ++ti) {
        // B operands of this k-tile (one LDS round trip),
        // then the A fragments of the next one
        const f32x4 b0 = xin[ti * 64], b1 = xin[(NT + ti) * 64];
        if (ti + 1 < NT) issue(l, ti + 1, (ti + 1) & 1);
        else issue(l_next, 0, 0);   // next layer (or the next tile's first layer)
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetches one k-tile ahead
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int to = 0; to < TO; ++to) {
            acc[0][to] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[ti & 1][to][r], b0[r], acc[0][to], 0, 0, 0);
            acc[1][to] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[ti & 1][to][r], b1[r], acc[1][to], 0, 0, 0);
          }
      }
      if (l + 1 < n_hidden) {
        float* xout = s_x + ((l + 1) & 1) * XBUF;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int to = 0; to < TO; ++to) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = vmc_act<ACT>(acc[hf][to][e]);
            *(f32x4*)(xout + ((hf * NT + wave * TO + to) * 64 + lane) * 4) = v;
          }
      } else {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          float part = 0.f;
#pragma unroll
          for (int to = 0; to < TO; ++to) {
            if (RBM) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                part += 16 * (wave * TO + to) + 4 * g + e < a.n_units ? vmc_logcosh(acc[hf][to][e]) : 0.f;
            } else {
              const f32x4 w = *(const f32x4*)(s_wout + 16 * (wave * TO + to) + 4 * g);
#pragma unroll
              for (int e = 0; e < 4; ++e) part = fmaf(vmc_act<ACT>(acc[hf][to][e]), w[e], part);
            }
          }
          part += __shfl_xor(part, 16);
          part += __shfl_xor(part, 32);
          if (g == 0) s_part[(wave * 2 + hf) * 16 + j] = part;
        }
      }
    }
    __syncthreads();
    if (wave == 0 && g < 2) {
      const int row = tile * 32 + 16 * g + j;
      float logit = ((s_part[(0 * 2 + g) * 16 + j] + s_part[(1 * 2 + g) * 16 + j]) +
                     (s_part[(2 * 2 + g) * 16 + j] + s_part[(3 * 2 + g) * 16 + j])) + bout;
      if (RBM) logit += s_meta[(16 * g + j) * 3 + 2];
      if (row < n_rows) {
        if (RATIO) a.out[row] = s_meta[(16 * g + j) * 3] * vmc_out_ratio(oact, logit, s_meta[(16 * g + j) * 3 + 1]);
        else a.out[row] = logit;
      }
    }
  }
}

inline size_t lds_bytes(int nt, int n_hidden) {
  return sizeof(float) * (size_t)(2 * (2 * nt * 256) + NW * 32 + 96 + n_hidden * nt * 16 + nt * 16);
}
}  // namespace tail_lds

template <int NT, bool RATIO, bool RBM, int ACT>
__global__ __launch_bounds__(256) void k_tail_lds(TailArgs a) { tail_lds::tail_lds_body<NT, RATIO, RBM, ACT>(a); }

template <int NT, bool RATIO, bool RBM, int ACT>
static hipError_t launch_tail_lds_t(hipStream_t s, const TailArgs& a) {
  const size_t lds = tail_lds::lds_bytes(NT, a.n_hidden);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const int tiles = (a.n_rows + 31) / 32;
  const int persistent = a.num_cus > 0 ? a.num_cus : 256;
  const dim3 grid(tiles < persistent ? tiles : persistent), block(256);
  hipError_t e = hipFuncSetAttribute((const void*)k_tail_lds<NT, RATIO, RBM, ACT>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_tail_lds<NT, RATIO, RBM, ACT>), grid, block, lds, s, a);
  return hipGetLastError();
}

// 384 or 512 padded units with at least one H x H layer
template <int ACT>
static hipError_t launch_tail_lds_act(hipStream_t s, const TailArgs& a, int Hp, bool ratio, bool rbm) {
  if (a.n_rows <= 0) return hipSuccess;
  if (a.n_hidden < 1) return hipErrorInvalidValue;
#define VMC_TL(NT)                                                                              \
  do {                                                                                          \
    if (rbm) return ratio ? launch_tail_lds_t<NT, true, true, ACT>(s, a) : launch_tail_lds_t<NT, false, true, ACT>(s, a); \
    return ratio ? launch_tail_lds_t<NT, true, false, ACT>(s, a) : launch_tail_lds_t<NT, false, false, ACT>(s, a);        \
  } while (0)
  if (Hp == 384) VMC_TL(24);
  if (Hp == 512) VMC_TL(32);
#undef VMC_TL
  return hipErrorInvalidValue;
}
