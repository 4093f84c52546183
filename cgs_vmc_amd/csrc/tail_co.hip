// k_tail_co: the local-energy row kernel that shares a CU with the sampler (fully_connected, relu,
// H = 256).  operators.py:137-169 / 249-259: one row per connected configuration, value
// 0.5 jx psi(row) / psi(chain).
//
// Status: measured, NOT the default (CGS_VMC_CO=1 selects it; DESIGN.md 5 has the numbers): the pair
// is resident together as designed, but the hardware arbitrates the matrix pipe per instruction, not
// per phase -- this kernel's MFMAs also delay the sampler's own MFMA phases, which are its critical
// path (s_setprio makes no difference), and the step comes out at 2.07 ms against 2.04 ms.
//
// Why it exists.  At config 3 (4096 chains) the sampler k_sweep16 has exactly one 16-chain tile per
// CU and its mc_step is one dependent chain, so the matrix pipe idles ~35 % of the sweep; k_tail16
// (402 registers, one wave per SIMD) cannot be resident next to it.  This kernel computes the same
// rows in 112 registers and half the LDS: with k_sweep16_co (2 x 200 registers per SIMD, W1 in L2)
// one workgroup of each fits a CU, and the hardware issues this kernel's MFMAs whenever the
// sampler's waves wait (proposals, z1' build, output dot, accept, barriers).
//
// Shape: 4 waves, 32 rows per tile (two 16-row halves that share every weight fragment).  Wave w
// owns output units 64 w .. 64 w + 63 (TO = 4 tiles) of both halves: 8 accumulator tiles; the B
// operands of a layer come from LDS (as in the sampler), the A fragments stream from L2 through a
// two-stage register ring (one k-tile = 4 KiB per wave = 32 MFMAs ahead).  One barrier per layer.
//
// The same body, without the register cap and for 24 or 32 unit tiles per row (384 / 512 hidden
// units: TO = 6 / 8 tiles per wave), is the row kernel of the wide fully_connected path
// (k_tail_lds; local energies and plain logits), where k_tail16's register-resident activations
// (2 x NT x 4 registers twice) no longer fit.
#include "common.hpp"

#ifndef TAILCO_VGPR
#define TAILCO_VGPR 56   // amdgpu_num_vgpr counts half of the unified file: 112 registers
#endif

namespace {
constexpr int NW = 4;

template <int NT, bool RATIO>
__device__ __forceinline__ void tail_lds_body(const TailArgs& a) {
  constexpr int Hp = NT * 16, TO = NT / NW;
  constexpr int XBUF = 2 * NT * 256;   // floats of one operand buffer [2 halves][NT][64 lanes][4]
  static_assert(NT % NW == 0 && TO % 2 == 0, "unit tiles divide over the waves in pairs");
  extern __shared__ float smem[];
  const int n_hidden = a.n_hidden;
  float* s_x = smem;                     // [2][2][NT][64][4]
  float* s_part = s_x + 2 * XBUF;        // [NW][2][16] partial output dots
  float* s_meta = s_part + NW * 32;      // [32][2] {0.5 jx of the row's bond, logit of its chain}
  float* s_bias = s_meta + 64;           // [n_hidden][Hp]
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
          s_meta[(16 * hf + j) * 2] = a.half_jx[bond];
          s_meta[(16 * hf + j) * 2 + 1] = a.logit_base[ri.x];
        }
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
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(coef, x[q][e] - y[q][e], z[q][e]), 0.f);
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
        // B operands of this k-tile (one LDS round trip, hidden by the co-resident sampler waves),
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
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[hf][to][e], 0.f);
            *(f32x4*)(xout + ((hf * NT + wave * TO + to) * 64 + lane) * 4) = v;
          }
      } else {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          float part = 0.f;
#pragma unroll
          for (int to = 0; to < TO; ++to) {
            const f32x4 w = *(const f32x4*)(s_wout + 16 * (wave * TO + to) + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) part = fmaf(fmaxf(acc[hf][to][e], 0.f), w[e], part);
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
      const float logit = ((s_part[(0 * 2 + g) * 16 + j] + s_part[(1 * 2 + g) * 16 + j]) +
                           (s_part[(2 * 2 + g) * 16 + j] + s_part[(3 * 2 + g) * 16 + j])) + bout;
      if (row < n_rows) {
        if (RATIO) a.out[row] = s_meta[(16 * g + j) * 2] * vmc_out_ratio(oact, logit, s_meta[(16 * g + j) * 2 + 1]);
        else a.out[row] = logit;
      }
    }
  }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(TAILCO_VGPR)))
void k_tail_co(TailArgs a) { tail_lds_body<16, true>(a); }

template <int NT, bool RATIO>
__global__ __launch_bounds__(256) void k_tail_lds(TailArgs a) { tail_lds_body<NT, RATIO>(a); }

size_t tail_lds_bytes_t(int nt, int n_hidden) {
  return sizeof(float) * (size_t)(2 * (2 * nt * 256) + NW * 32 + 64 + n_hidden * nt * 16 + nt * 16);
}

template <int NT, bool RATIO>
hipError_t launch_tail_lds_t(hipStream_t s, const TailArgs& a) {
  const size_t lds = tail_lds_bytes_t(NT, a.n_hidden);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const int tiles = (a.n_rows + 31) / 32;
  const int persistent = a.num_cus > 0 ? a.num_cus : 256;
  const dim3 grid(tiles < persistent ? tiles : persistent), block(256);
  hipError_t e = hipFuncSetAttribute((const void*)k_tail_lds<NT, RATIO>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_tail_lds<NT, RATIO>), grid, block, lds, s, a);
  return hipGetLastError();
}
}  // namespace

size_t tail_co_lds_bytes(int n_hidden) { return tail_lds_bytes_t(16, n_hidden); }

// lds_bytes >= tail_co_lds_bytes(n_hidden): the caller pads it beyond half a CU's LDS so that two
// of these workgroups never share a CU (the second one would take the sampler's place)
hipError_t launch_tail_co(hipStream_t s, const TailArgs& a, size_t lds_bytes) {
  if (a.n_rows <= 0) return hipSuccess;
  if (a.n_hidden < 1 || lds_bytes < tail_co_lds_bytes(a.n_hidden) || lds_bytes > 160 * 1024)
    return hipErrorInvalidValue;
  const int tiles = (a.n_rows + 31) / 32;
  const int persistent = a.num_cus > 0 ? a.num_cus : 256;
  const dim3 grid(tiles < persistent ? tiles : persistent), block(256);
  hipError_t e = hipFuncSetAttribute((const void*)k_tail_co,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_tail_co, grid, block, lds_bytes, s, a);
  return hipGetLastError();
}

// fully_connected (relu) with 384 or 512 padded units and at least one H x H layer: 0.5 jx psi'/psi
// of the rows of a row list (ratio) or their logits
bool tail_lds_supported(int Hp, int n_hidden) {
  return (Hp == 384 || Hp == 512) && n_hidden >= 1 && tail_lds_bytes_t(Hp / 16, n_hidden) <= 160 * 1024;
}

hipError_t launch_tail_lds(hipStream_t s, const TailArgs& a, int Hp, bool ratio) {
  if (a.n_rows <= 0) return hipSuccess;
  if (!tail_lds_supported(Hp, a.n_hidden)) return hipErrorInvalidValue;
  if (Hp == 384) return ratio ? launch_tail_lds_t<24, true>(s, a) : launch_tail_lds_t<24, false>(s, a);
  return ratio ? launch_tail_lds_t<32, true>(s, a) : launch_tail_lds_t<32, false>(s, a);
}
