// k_cgen_band: one periodic convolution of the GENERAL convolution path (conv_general.hip: feature maps in HBM,
// channel-last [row][site][Fp]) at up to 64 filters, as an implicit GEMM on bands of lattice rows staged through LDS
// (round 6, VERDICT r5 "missing 2" / item 3a: lattices whose maps exceed the LDS at few filters -- the fused kernels of
// conv_kernels.hpp keep a sample's two maps in LDS and refuse them; the im2col + GEMM form of this path wrote a
// [rows N][taps F] matrix per convolution and multiplied it into 16 of a tile's 64 columns: 0.04-0.065 of the fp32-MFMA
// peak at 36 x 36 sites x 16 filters).  The reference takes any lattice: layers.py:89-160, wavefunctions.py:534-579.
//
// The formulation is the fused kernels': the output tile of one v_mfma_f32_16x16x4_f32 is 16 output channels x 16
// lattice positions, reduced over (tap, input channel) four channels at a time; A = a weight fragment (all K KW 4 of a
// layer stay in registers for the whole launch), B = one ds_read_b128 of the staged input per tap (lane (p, g): channels
// 4g .. 4g+3 of the site `tap` away from position p); the accumulator has the position on the lane and channel 4g + r on
// register r: one 16-byte store per lane.  A persistent workgroup (4 waves) walks the items (row, band of BH lattice rows);
// per item it stages the band WITH its periodic halo -- (BH + K - 1) x (D2 + KW - 1) sites x 16 channels, the wrap of
// layers.py:118-148 resolved while staging -- so that a tap is a constant LDS offset and a tile may run across lattice
// rows; several workgroups per CU hide each other's staging.  The first convolution (one input channel: the spins, the
// exchanged pair of a connected configuration / proposed move negated while staging, operators.py:162-163,
// graph_builders.py:67-71) runs the taps over the MFMA's k index, four per instruction.
// HBM traffic per convolution and row: the input map (x (BH + K - 1) / BH) in, the output map out -- no im2col matrix.
// Same arithmetic as the fused kernels / k_gemm up to the order of additions (taps outer, channels inner, fp32 MFMA).
#include "conv.hpp"

#include <type_traits>

namespace {

__device__ __forceinline__ float cb_selu(float x) {   // layers.py:226 tf.nn.selu (constants of conv_general.hip)
  const float scale = 1.0507009873554805f, alpha = 1.6732632423543772f;
  return scale * (x > 0.f ? x : alpha * (expf(x) - 1.f));
}
__device__ __forceinline__ float cb_pre(int pre, float x) {
  return pre < 0 ? x : (pre == CGEN_PRE_SELU ? cb_selu(x) : vmc_act_rt(pre, x));
}
__device__ __forceinline__ int cb_wrap(int v, int d) { v %= d; return v < 0 ? v + d : v; }

// NCB: channel blocks of 16 (filters <= 16 NCB).  One block: 4 waves, several workgroups per CU.  More: ONE workgroup per
// CU of NCB x WPC waves (8, 6, 8 at 2, 3, 4 blocks) -- WPC waves per OUTPUT block `co`, each holding that block's fragments
// against every input block (K KW 4 NCB registers) for the whole launch -- which stages an item's band ONCE, into up to
// 144 KB of LDS, for all of its output blocks (the first form had a workgroup per output block: every band was staged NCB
// times, in thin bands with a (K - 1)-row halo each: the L2 -> LDS copies, not the products, bounded it).  The staged band is
// block-major, [input block][site][16 channels]: consecutive positions are 64 contiguous bytes in every block
// (channel-last with 64 NCB bytes per site would put eight positions on the same banks).
template <int NCB> struct BandShape {
  static constexpr int WPC = NCB == 1 ? 4 : (NCB == 2 ? 4 : 2);     // waves per output block
  static constexpr int NWV = NCB == 1 ? 4 : NCB * WPC;              // waves per workgroup
};
template <int K, int KW, int NCB, bool FIRST>
__global__ __launch_bounds__(64 * BandShape<NCB>::NWV, (NCB > 1 ? 1 : (K * KW <= 9 ? 4 : (K * KW <= 50 ? 2 : 1)))) void k_cgen_band(CgenBandArgs a) {
  constexpr int T = K * KW;
  constexpr int NF = FIRST ? (T + 3) / 4 : NCB * T * 4;   // weight fragments (one VGPR each)
  constexpr int WPC = BandShape<NCB>::WPC, NTHR = 64 * BandShape<NCB>::NWV;
  extern __shared__ float s_band[];                       // FIRST: [SR][SC]; else [NCB][SR][SC][16]
  const ConvGeom g = a.g;
  const int D1 = g.D1, D2 = g.D2, N = g.N, F = g.F, Fp = a.Fp;
  const int BH = a.band_rows, SC = D2 + KW - 1, NB = (D1 + BH - 1) / BH;
  const int plane = (BH + K - 1) * SC * 16;               // floats of one input block of the staged band
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 15, gq = lane >> 4;
  const int co = NCB > 1 ? wave / WPC : 0;                // this wave's output block
  const int wl = NCB > 1 ? wave % WPC : wave;             // ... and its place among that block's waves
  const int fo = 16 * co + p;                             // output channel of this lane's A rows

  // ---- weight fragments (A operand: lane (cout = fo, k slot gq))
  float wf[NF];
  if (FIRST) {        // w[tap][0][F]: k index = tap 4 m + gq
#pragma unroll
    for (int m = 0; m < NF; ++m) {
      const int t = 4 * m + gq;
      wf[m] = (t < T && fo < F) ? a.w[(long long)t * F + fo] : 0.f;
    }
  } else {            // w[tap][cin][F]: MFMA e of (input block ci, tap) contracts channels 16 ci + 4 gq + e
#pragma unroll
    for (int ci = 0; ci < NCB; ++ci)
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c_in = 16 * ci + 4 * gq + e;
          wf[(ci * T + t) * 4 + e] = (c_in < F && fo < F) ? a.w[((long long)t * F + c_in) * F + fo] : 0.f;
        }
  }
  f32x4 bias4;
#pragma unroll
  for (int r = 0; r < 4; ++r) bias4[r] = 16 * co + 4 * gq + r < F ? a.bias[16 * co + 4 * gq + r] : 0.f;
  // first convolution: LDS offset of this lane's tap in MFMA m
  int toff[FIRST ? NF : 1];
  if (FIRST) {
#pragma unroll
    for (int m = 0; m < NF; ++m) {
      const int t = 4 * m + gq, tt = t < T ? t : 0;
      toff[m] = (tt / KW) * SC + (tt % KW);
    }
  }

  const long long n_items = (long long)a.rows * NB;
  for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int r = (int)(item / NB), b = (int)(item - (long long)r * NB);
    const int y0 = b * BH, bh = min(BH, D1 - y0);
    __syncthreads();                 // the previous item's readers are done with the band
    // ---- stage the band with its periodic halo
    if (FIRST) {
      int chain = (int)a.row0 + r, fa = -1, fb = -1;
      if (a.rowinfo) {
        const int2 ri = a.rowinfo[a.row0 + r];
        chain = ri.x;
        if (ri.y != 0) { const int2 ab = a.bonds[(ri.y > 0 ? ri.y : -ri.y) - 1]; fa = ab.x; fb = ab.y; }
      }
      if (a.iup) { fa = a.iup[a.row0 + r]; fb = a.idn[a.row0 + r]; }
      const float* src = a.configs + (long long)chain * N;
      const int total = (bh + K - 1) * SC;
      for (int i0 = tid; i0 < total; i0 += 4 * NTHR) {      // four loads in flight per thread
        float xv[4]; int sv[4];
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
          const int i = min(i0 + uu * NTHR, total - 1);
          const int sy = i / SC, sx = i - sy * SC;
          sv[uu] = cb_wrap(y0 + sy - g.lo, D1) * D2 + cb_wrap(sx - g.lo2, D2);
          xv[uu] = src[sv[uu]];
        }
#pragma unroll
        for (int uu = 0; uu < 4; ++uu)
          if (i0 + uu * NTHR < total) s_band[i0 + uu * NTHR] = (sv[uu] == fa || sv[uu] == fb) ? -xv[uu] : xv[uu];
      }
    } else {
      // eight 16-byte loads in flight per thread (a load -> store loop waits one HBM round trip per piece: the staging,
      // not the products, bounded the first form of this kernel); addresses clamped, stores masked
      const float* src = a.in + (long long)r * N * Fp;
      const int total = (bh + K - 1) * SC * 4 * NCB;
      constexpr int SB = NF > 150 ? 2 : 8;     // loads in flight per thread (fewer where the fragments leave few registers)
      for (int i0 = tid; i0 < total; i0 += SB * NTHR) {
        f32x4 v[SB];
        int dsto[SB], ch[SB];
#pragma unroll
        for (int uu = 0; uu < SB; ++uu) {
          const int i = min(i0 + uu * NTHR, total - 1);
          const int site = i / (4 * NCB), cq = i - site * (4 * NCB);       // cq: 4-channel quad of the site, 0 .. 4 NCB - 1
          const int sy = site / SC, sx = site - sy * SC;
          const int sl = cb_wrap(y0 + sy - g.lo, D1) * D2 + cb_wrap(sx - g.lo2, D2);
          v[uu] = *(const f32x4*)(src + (long long)sl * Fp + (4 * cq < Fp ? 4 * cq : 0));
          dsto[uu] = (cq >> 2) * plane + site * 16 + 4 * (cq & 3);
          ch[uu] = 4 * cq;                                                 // first channel of the quad (>= F: zeros)
        }
#pragma unroll
        for (int uu = 0; uu < SB; ++uu) {
          if (i0 + uu * NTHR < total) {
            f32x4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = ch[uu] + e < F ? cb_pre(a.pre_act, v[uu][e]) : 0.f;
            *(f32x4*)(s_band + dsto[uu]) = w;
          }
        }
      }
    }
    __syncthreads();
    // ---- position tiles of the band: 16 consecutive positions (row-major over the band's bh x D2 sites)
    const int n_pos = bh * D2, n_tiles = (n_pos + 15) >> 4;
    for (int tile = wl; tile < n_tiles; tile += WPC) {
      const int q = tile * 16 + p, qq = q < n_pos ? q : n_pos - 1;      // (a ragged last tile computes its last position again)
      const int y = qq / D2, x = qq - y * D2;
      f32x4 acc = bias4;
      if (FIRST) {
        const float* base = s_band + y * SC + x;
#pragma unroll
        for (int m = 0; m < NF; ++m)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[m], base[toff[m]], acc, 0, 0, 0);
      } else {
        const float* base = s_band + (y * SC + x) * 16 + 4 * gq;
#pragma unroll
        for (int ci = 0; ci < NCB; ++ci)
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const f32x4 bv = *(const f32x4*)(base + ci * plane + ((t / KW) * SC + (t % KW)) * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[(ci * T + t) * 4 + e], bv[e], acc, 0, 0, 0);
          }
      }
      asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc));    // the relu of vmc_act_rt is an asm v_max_f32 (common.hpp: vmc_mfma_settle)
      // ---- epilogue (GemmArgs' ids): 1 f(v + bias), 4 v + bias, 8 C + v + bias, 11 selu(v + bias); bias is in acc
      const int ch0 = 16 * co + 4 * gq;
      if (q < n_pos && ch0 < Fp) {
        float* dst = a.out + ((long long)r * N + (long long)(y0 + y) * D2 + x) * Fp + ch0;
        f32x4 v = acc;
        if (a.epilogue == 8) {
          const f32x4 c = *(const f32x4*)dst;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += c[e];
        } else if (a.epilogue == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = vmc_act_rt(a.act, v[e]);
        } else if (a.epilogue == 11) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = cb_selu(v[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ch0 + e < F ? v[e] : 0.f;
        *(f32x4*)dst = v;
      }
    }
  }
}

// First convolution (one input channel) at MORE than 16 filters, directly: out[row][n][f] = epilogue(b[f] + sum_t w[t][f] s'(site(n, t)))
// with s' = the row's spins, the exchanged pair negated -- instead of an im2col matrix [rows N][taps] and a K = taps GEMM
// (48 + 15 us per mc_step of 0.7 ms at 3 x 128 filters on 10 x 10).  One workgroup per row at a time; the spins, the
// weights and the neighbour table live in LDS; a thread makes four channels of a site (taps x 4 fused multiply-adds) and
// stores 16 bytes: the launch is bounded by the write of the map.  Taps ascending, as the im2col + GEMM form's k index.
__global__ __launch_bounds__(256) void k_cgen_first_direct(CgenBandArgs a) {
  extern __shared__ float s_first[];
  const ConvGeom g = a.g;
  const int N = g.N, T = g.K * g.KW, F = g.F, Fp = a.Fp, FQ = Fp / 4;
  float* s_spin = s_first;                 // [N]
  float* s_w = s_spin + ((N + 3) & ~3);    // [T][Fp]
  float* s_b = s_w + T * Fp;               // [Fp]
  int* s_nb = (int*)(s_b + Fp);            // [N][T]
  const int tid = threadIdx.x;
  for (int i = tid; i < T * Fp; i += 256) { const int t = i / Fp, f = i - t * Fp; s_w[i] = f < F ? a.w[(long long)t * F + f] : 0.f; }
  for (int i = tid; i < Fp; i += 256) s_b[i] = i < F ? a.bias[i] : 0.f;
  for (int i = tid; i < N * T; i += 256) {
    const int n = i / T, t = i - n * T;
    const int a1 = n / g.D2, a2 = n - a1 * g.D2, d1 = t / g.KW, d2 = t - d1 * g.KW;
    s_nb[i] = cb_wrap(a1 + d1 - g.lo, g.D1) * g.D2 + cb_wrap(a2 + d2 - g.lo2, g.D2);
  }
  for (int r = blockIdx.x; r < a.rows; r += gridDim.x) {
    int chain = (int)a.row0 + r, fa = -1, fb = -1;
    if (a.rowinfo) {
      const int2 ri = a.rowinfo[a.row0 + r];
      chain = ri.x;
      if (ri.y != 0) { const int2 ab = a.bonds[(ri.y > 0 ? ri.y : -ri.y) - 1]; fa = ab.x; fb = ab.y; }
    }
    if (a.iup) { fa = a.iup[a.row0 + r]; fb = a.idn[a.row0 + r]; }
    __syncthreads();
    for (int i = tid; i < N; i += 256) { const float x = a.configs[(long long)chain * N + i]; s_spin[i] = (i == fa || i == fb) ? -x : x; }
    __syncthreads();
    float* out = a.out + (long long)r * N * Fp;
    for (int i = tid; i < N * FQ; i += 256) {
      const int n = i / FQ, cq = i - n * FQ;
      f32x4 v = *(const f32x4*)(s_b + 4 * cq);
      const int* nb = s_nb + n * T;
      for (int t = 0; t < T; ++t) {
        const float x = s_spin[nb[t]];
        const f32x4 w = *(const f32x4*)(s_w + t * Fp + 4 * cq);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(x, w[e], v[e]);
      }
      if (a.epilogue == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = vmc_act_rt(a.act, v[e]);
      } else if (a.epilogue == 11) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = cb_selu(v[e]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = 4 * cq + e < F ? v[e] : 0.f;
      *(f32x4*)(out + (long long)n * Fp + 4 * cq) = v;
    }
  }
}

template <int K, int KW, int NCB, bool FI>
hipError_t launch_kf(hipStream_t s, CgenBandArgs a, int num_cus) {
  constexpr int NTHR = 64 * BandShape<NCB>::NWV;
  // workgroups per CU: what the runtime says the kernel's registers and the widest band's LDS allow (at most 4; more
  // than one channel block: one workgroup of 6 or 8 waves per CU)
  const size_t lds_max = plan_cgen_band_lds_bytes(a.g, FI);
  hipError_t e = hipFuncSetAttribute((const void*)k_cgen_band<K, KW, NCB, FI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
  if (e != hipSuccess) return e;
  int per_cu = 1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_cgen_band<K, KW, NCB, FI>, NTHR, lds_max) != hipSuccess || per_cu < 1)
    per_cu = 1;
  if (per_cu > 4) per_cu = 4;
  a.band_rows = plan_cgen_band_rows_for(a.g, a.rows, (long long)num_cus * per_cu);
  const size_t lds = plan_cgen_band_lds_bytes(a.g, FI, a.band_rows);
  const int NB = (a.g.D1 + a.band_rows - 1) / a.band_rows;
  const long long items = (long long)a.rows * NB;
  long long grid = (long long)num_cus * per_cu;
  if (grid > items) grid = items;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((k_cgen_band<K, KW, NCB, FI>), dim3((unsigned)grid), dim3(NTHR), lds, s, a);
  return hipGetLastError();
}
template <int K, int KW, int NCB>
hipError_t launch_k(hipStream_t s, const CgenBandArgs& a, int num_cus) {
  if (a.layer == 0) return launch_kf<K, KW, NCB, true>(s, a, num_cus);
  return launch_kf<K, KW, NCB, false>(s, a, num_cus);
}

// NCB by the filter count; the shapes whose fragments fit the registers: plan_cgen_band_ok
template <int K, int KW>
hipError_t launch_kk(hipStream_t s, const CgenBandArgs& a, int num_cus) {
  const int ncb = (a.g.F + 15) / 16;
  constexpr int T = K * KW;
  if (ncb == 1) return launch_k<K, KW, 1>(s, a, num_cus);
  if constexpr (T * 2 <= PLAN_CGEN_BAND_MAX_FRAGS / 4) { if (ncb == 2) return launch_k<K, KW, 2>(s, a, num_cus); }
  if constexpr (T * 3 <= PLAN_CGEN_BAND_MAX_FRAGS / 4) { if (ncb == 3) return launch_k<K, KW, 3>(s, a, num_cus); }
  if constexpr (T * 4 <= PLAN_CGEN_BAND_MAX_FRAGS / 4) { if (ncb == 4) return launch_k<K, KW, 4>(s, a, num_cus); }
  return hipErrorInvalidValue;
}

}  // namespace

// (rows per band and the shapes taken: plan_cgen_band_rows / plan_cgen_band_ok, plan.hpp -- pure host, under ASan in hostcheck)
int cgen_band_rows(const ConvGeom& g) { return plan_cgen_band_rows(g); }
bool cgen_band_ok(const ConvGeom& g) { return plan_cgen_band_ok(g); }

// the first convolution of any filter count when the neighbour table fits LDS (epilogues 1 / 4 / 11: no residual add in front)
bool cgen_first_direct_ok(const ConvGeom& g, int epilogue) {
  return (epilogue == 1 || epilogue == 4 || epilogue == 11) && plan_cgen_first_direct_lds_bytes(g) <= 64 * 1024;
}
hipError_t launch_cgen_first_direct(hipStream_t s, const CgenBandArgs& a, int num_cus) {
  if (a.rows <= 0) return hipSuccess;
  if (a.layer != 0 || !cgen_first_direct_ok(a.g, a.epilogue)) return hipErrorInvalidValue;
  const size_t lds = plan_cgen_first_direct_lds_bytes(a.g);
  hipError_t e = hipFuncSetAttribute((const void*)k_cgen_first_direct, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  long long grid = (long long)num_cus * 2;
  if (grid > a.rows) grid = a.rows;
  hipLaunchKernelGGL(k_cgen_first_direct, dim3((unsigned)grid), dim3(256), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_cgen_band(hipStream_t s, const CgenBandArgs& a_in, int num_cus) {
  if (a_in.rows <= 0) return hipSuccess;
  if (!cgen_band_ok(a_in.g)) return hipErrorInvalidValue;
  const CgenBandArgs& a = a_in;
  const bool two_d = a.g.KW == a.g.K;
#define CB_CASE(KK) case KK: return two_d ? launch_kk<KK, KK>(s, a, num_cus) : launch_kk<KK, 1>(s, a, num_cus);
  switch (a.g.K) {
    CB_CASE(2) CB_CASE(3) CB_CASE(4) CB_CASE(5) CB_CASE(6) CB_CASE(7)
    default: return hipErrorInvalidValue;
  }
#undef CB_CASE
}
