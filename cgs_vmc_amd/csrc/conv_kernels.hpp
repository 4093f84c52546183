// Convolutional ansatz kernels (Conv2DNetwork / ResNet2D, wavefunctions.py:531-615, 710-809, and
// their 1-D siblings Conv1DNetwork / ResNet1D, 455-527, 618-707, as k x 1 taps on an N x 1 lattice)
// for gfx950, templated on (kernel size K, taps along axis 2 KW, channel blocks NCB).  Included by
// conv.hip (NCB = 1: up to 16 filters) and, through conv_wide.hpp, conv32 / conv48 / conv64.hip (NCB = 2, 3, 4:
// 17 .. 64 filters).
// See DESIGN.md 4 "Convolutional ansatz types".
//
// A periodic convolution with <= 16 channels is an implicit GEMM whose output tile is exactly one
// v_mfma_f32_16x16x4_f32 tile: 16 output channels x 16 lattice positions, reduced over
// (tap, input channel) four input channels at a time.  The A operand is a weight fragment (all
// K*K*4 of them stay in registers for the whole layer), the B operand is one ds_read_b128 of the
// input feature map per tap: lane (p, g) reads channels 4g..4g+3 of the site `tap` away from
// position p.  The accumulator comes out with the position on the lane and channel 4g+r on
// register r, which is the layout the next layer reads, so the epilogue is one ds_write_b128.
// Feature maps of the G samples a workgroup has in flight never leave LDS between layers.
//
// More than 16 filters (NCB = 2 .. 4 channel blocks of 16): a convolution is NCB x NCB such block
// products.  The fragments of ONE (output block, input block) pair are held in registers -- in
// chunks of at most 25 taps -- while the wave sweeps its position tiles, whose accumulators stay in
// registers across the chunks and the input blocks: the weights are re-read from L2 once per block
// pair and layer -- the same bytes per MFMA as the single-block kernel.
#pragma once
#include "conv.hpp"
#include <cstdlib>
#include <type_traits>

// NCB = 1 (conv.hip): 4 waves per workgroup (one per SIMD) and two workgroups per CU: the two
// co-resident workgroups are never in step, so the serial phases of one (row staging behind dependent
// global loads, the per-layer weight-fragment reload, barriers, the final reduction) run under the
// MFMAs of the other (one 4-wave workgroup per CU: 0.58 of the fp32-MFMA peak against 0.72 for two).
// NCB >= 2 (conv_wide.hpp defines CONV_WAVES 8): a sample's feature maps are at least twice as large, only two
// samples fit half a CU's LDS and their 13 position tiles divide badly over 4 waves; one 8-wave
// workgroup per CU with the whole 160 KiB (five samples on a 10 x 10 lattice: 16 tile pairs, two per
// wave) keeps two waves per SIMD and every wave busy.
#ifndef CONV_WAVES
#define CONV_WAVES 4
#endif
#define CONV_WG_PER_CU (CONV_WAVES == 4 ? 2 : 1)
#define CONV_THREADS (CONV_WAVES * 64)
// (CONV_LDS_PER_WG, the 80 KiB a workgroup gets when two share a CU: plan.hpp)
#define SELU_SCALE_F 1.0507009873554805f
#define SELU_ALPHA_F 1.6732632423543772f

namespace {

__device__ __forceinline__ float selu_f(float x) {
  return SELU_SCALE_F * (x > 0.f ? x : SELU_ALPHA_F * (expf(x) - 1.f));
}
// selu'(u) from t = selu(u): scale for u > 0, else scale * alpha * e^u = t + scale * alpha
__device__ __forceinline__ float selu_deriv_from_t(float t) {
  return t > 0.f ? SELU_SCALE_F : t + SELU_SCALE_F * SELU_ALPHA_F;
}

// f'(z) from the taped value (run-time activation id): a = f(z) for every activation whose
// derivative is a function of a; the cosine tapes z itself (f' = -sin z)
__device__ __forceinline__ float dact_from_tape_rt(int act, float a) {
  switch (act) {
    case VMC_ACT_RELU_: return a > 0.f ? 1.f : 0.f;
    case VMC_ACT_EXP_: return a;
    case VMC_ACT_COS_: return -__sinf(a);
    case VMC_ACT_TAN_: return 1.f + a * a;
    case VMC_ACT_TANH_: return 1.f - a * a;
    case VMC_ACT_SIGMOID_: return a * (1.f - a);
    default: return 1.f;
  }
}

// epilogues of one convolution over the LDS-resident samples
enum { EP_LINEAR = 0,     // out = acc + bias
       EP_ACT = 1,        // out = f(acc + bias), f = hidden activation id
       EP_SELU = 2,       // out = selu(acc + bias)
       EP_RESADD = 3,     // out = out + acc + bias              (ResBlock2d shortcut, layers.py:228)
       EP_BACK_DACT = 4,  // out = acc * f'(tape)                (back-propagation, conv_2d)
       EP_BACK_SELU = 5,  // out = acc * selu'(tape)
       EP_BACK_ADD = 6,   // out = out + acc
       EP_DOT = 7 };      // out = out + (acc + bias) * delta(tape_in)   (SR: O_b . p, summed per sample later)

// per-position descriptor: sample slot, lattice coordinates (built once per kernel, LDS)
__device__ __forceinline__ unsigned pack_pos(int s, int a1, int a2) {
  return (unsigned)a2 | ((unsigned)a1 << 10) | ((unsigned)s << 20);
}

struct ConvSmem {
  float* xs;        // [G][XS] spins
  float* buf0;      // [G][CS]
  float* buf1;      // [G][CS]
  unsigned* pinfo;  // [G N] pack_pos
  int* row_chain;   // [G] chain (or row) of each slot, -1 = empty
  float* red;       // [G] reduced logits
  // periodic neighbour tables, built once per kernel (the wrap arithmetic costs ~15 VALU
  // instructions per tap column / row and tile otherwise): rtab[dir][a1][d] = 16 D2 ((a1 + d - lo)
  // mod D1), ctab[dir][a2][d] = 16 ((a2 + d - lo) mod D2) in bytes (one v_add3 per tap and tile), lo = g.lo (dir 0: forward) or
  // g.hi (dir 1: transposed convolution); rows of plan_conv_tab(g) ints (8; 16 beyond 8 taps per axis)
  int* rtab;        // [2][D1][tab]
  int* ctab;        // [2][D2][tab]
};

// f(integral_constant<int, 0>), ..., f(integral_constant<int, N - 1>)
template <typename F, int... I>
__device__ __forceinline__ void conv_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void conv_static_for(F&& f) {
  conv_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ int conv_xs_stride(const ConvGeom& g) { return (g.N + 3) & ~3; }

__device__ __forceinline__ ConvSmem conv_carve(float* base, const ConvGeom& g, int G) {
  ConvSmem s;
  s.buf0 = base;
  s.buf1 = s.buf0 + (size_t)G * g.CS;
  s.xs = s.buf1 + (size_t)G * g.CS;
  s.pinfo = (unsigned*)(s.xs + (size_t)G * conv_xs_stride(g));
  s.row_chain = (int*)(s.pinfo + (size_t)G * g.N);
  s.red = (float*)(s.row_chain + G);
  s.rtab = (int*)(s.red + 6 * G);       // red, cur_logit, prop[2], prop_u of the sampler + spare
  s.ctab = s.rtab + 2 * g.D1 * plan_conv_tab(g);
  return s;
}

__device__ __forceinline__ int wrap(int v, int d) {
  v += v < 0 ? d : 0;
  v -= v >= d ? d : 0;
  return v;
}

__device__ __forceinline__ void conv_build_pinfo(const ConvSmem& sm, const ConvGeom& g, int G) {
  for (int q = threadIdx.x; q < G * g.N; q += blockDim.x) {
    const int s = q / g.N, site = q - s * g.N;
    const int a1 = site / g.D2, a2 = site - a1 * g.D2;
    sm.pinfo[q] = pack_pos(s, a1, a2);
  }
  const int tab = plan_conv_tab(g);
  for (int i = threadIdx.x; i < 2 * (g.D1 + g.D2) * tab; i += blockDim.x) {
    const bool is_r = i < 2 * g.D1 * tab;
    const int j = is_r ? i : i - 2 * g.D1 * tab, D = is_r ? g.D1 : g.D2;
    const int dir = j / (D * tab), a = (j / tab) % D, d = j & (tab - 1);
    const int lo = is_r ? (dir ? g.hi : g.lo) : (dir ? g.hi2 : g.lo2);
    const int w = ((a + min(d, (is_r ? g.K : g.KW) - 1) - lo) % D + D) % D;
    (is_r ? sm.rtab : sm.ctab)[j] = is_r ? 16 * g.D2 * w : 16 * w;     // byte offsets
  }
}

// activation epilogue + tape value of a forward convolution: what goes to the next layer and what
// the gradient path reads back (the cosine tapes the pre-activation)
__device__ __forceinline__ void conv_act_epilogue(int ep, int hact, f32x4& v, f32x4& taped) {
  if (ep == EP_ACT) {
    const f32x4 z = v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = vmc_act_rt(hact, v[r]);
    taped = hact == VMC_ACT_COS_ ? z : v;
  } else if (ep == EP_SELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = selu_f(v[r]);
    taped = v;
  } else {
    taped = v;
  }
}

// First convolution (one input channel, layers.py:151-160 on the reshaped spins): the k index of
// the MFMA runs over the taps, four per instruction.
template <int K, int KW, int NCB>
__device__ __forceinline__ void conv_first(const ConvSmem& sm, float* out, const ConvGeom& g,
                                           const ConvParams& p, int G, int ep, int wave, int lane,
                                           float* tape_out, long long tape_rows) {
  constexpr int Q0 = (K * KW + 3) / 4;
  constexpr int CONV_TAB = K <= 8 ? 8 : 16;       // plan_conv_tab(g)
  const int pl = lane & 15, gl = lane >> 4;
  float w0[NCB][Q0];
  f32x4 bias[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
    for (int q = 0; q < Q0; ++q) w0[cb][q] = p.w0[(cb * Q0 + q) * 64 + lane];
    bias[cb] = *(const f32x4*)(p.bias + 16 * cb + 4 * gl);
  }
  const int n_pos = G * g.N, n_tiles = (n_pos + 15) >> 4;
  const int xs_stride = conv_xs_stride(g);
  int d1[Q0], d2[Q0];
#pragma unroll
  for (int q = 0; q < Q0; ++q) {
    int tap = 4 * q + gl;
    tap = tap < K * KW ? tap : 0;         // the weight of a tap beyond K*KW is zero
    d1[q] = tap / KW;
    d2[q] = tap % KW;
  }
  for (int t = wave; t < n_tiles; t += CONV_WAVES) {
    const int q = t * 16 + pl;
    const bool valid = q < n_pos;
    const unsigned info = sm.pinfo[valid ? q : n_pos - 1];
    const int a2 = info & 1023, a1 = (info >> 10) & 1023, s = info >> 20;
    const float* xs = sm.xs + s * xs_stride;
    float bx[Q0];
#pragma unroll
    for (int qq = 0; qq < Q0; ++qq) bx[qq] = xs[(sm.rtab[a1 * CONV_TAB + d1[qq]] + sm.ctab[a2 * CONV_TAB + d2[qq]]) >> 4];
    const int site = a1 * g.D2 + a2;
    const int row = sm.row_chain[s];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      f32x4 acc = bias[cb];
#pragma unroll
      for (int qq = 0; qq < Q0; ++qq) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[cb][qq], bx[qq], acc, 0, 0, 0);
      f32x4 taped;
      float* dst = out + (size_t)s * g.CS + (4 * cb + gl) * g.GS + 4 * site;
      if (ep == EP_DOT) {      // tape_out is delta_0 here: out += (conv + bias) * delta
        if (valid) {
          const f32x4 d = row >= 0 ? *(const f32x4*)(tape_out + ((long long)row * 4 * NCB + 4 * cb + gl) * g.GS + 4 * site)
                                   : f32x4{0.f, 0.f, 0.f, 0.f};
          *(f32x4*)dst = *(const f32x4*)dst + acc * d;
        }
        continue;
      }
      conv_act_epilogue(ep, g.hact, acc, taped);
      if (valid) {
        *(f32x4*)dst = acc;
        if (tape_out && row >= 0)
          *(f32x4*)(tape_out + ((long long)row * 4 * NCB + 4 * cb + gl) * g.GS + 4 * site) = taped;
      }
    }
  }
}

// epilogue of one output tile of conv_layer: activation / residual add / derivative, LDS + tape
__device__ __forceinline__ void conv_store_tile(const ConvSmem& sm, const ConvGeom& g, float* out, int ep, int ncb,
                                                int co, int gl, bool valid, int sl, int site, f32x4 v,
                                                const float* tape_in, float* tape_out) {
  float* dst = out + (size_t)sl * g.CS + (4 * co + gl) * g.GS + 4 * site;
  const int row = sm.row_chain[sl];
  const unsigned trow = (unsigned)((row >= 0 ? row : 0) * 4 * ncb + 4 * co + gl) * (unsigned)g.GS + 4u * site;   // < 2^31: checked on the host
  f32x4 taped = v;
  if (ep == EP_ACT || ep == EP_SELU) {
    conv_act_epilogue(ep, g.hact, v, taped);
  } else if (ep == EP_RESADD || ep == EP_BACK_ADD) {
    const f32x4 old = *(const f32x4*)dst;
    v += old;
    taped = v;
  } else if (ep == EP_BACK_DACT || ep == EP_BACK_SELU) {
    const f32x4 a = *(const f32x4*)(tape_in + trow);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      v[r] *= ep == EP_BACK_SELU ? selu_deriv_from_t(a[r]) : dact_from_tape_rt(g.hact, a[r]);
    taped = v;
  } else if (ep == EP_DOT) {
    const f32x4 d = row >= 0 ? *(const f32x4*)(tape_in + trow) : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 old = *(const f32x4*)dst;
    v = old + v * d;
  }
  if (valid) {
    *(f32x4*)dst = v;
    if (tape_out && row >= 0) *(f32x4*)(tape_out + trow) = taped;
  }
}

// One convolution over the G resident samples: in -> out (LDS).  `wfrag` is the layer's fragment
// image ([NCB co][NCB ci][K*KW][64] f32x4, forward or transposed), `dir` selects the padding.
template <int K, int KW, int NCB>
__device__ __forceinline__ void conv_layer(const ConvSmem& sm, const float* in, float* out,
                                           const ConvGeom& g, const float* wfrag, const float* bias16,
                                           int dir, int G, int ep, int wave, int lane,
                                           const float* tape_in, float* tape_out) {
  constexpr int KK = K * KW;
  constexpr int CONV_TAB = K <= 8 ? 8 : 16;       // plan_conv_tab(g)
  const int pl = lane & 15, gl = lane >> 4;
  const int n_pos = G * g.N, n_tiles = (n_pos + 15) >> 4;
  // position descriptors + the tap loop of NTL adjacent tiles for one (co, ci) block pair:
  // independent accumulator chains that share every weight fragment.  B operands two taps ahead of
  // the MFMAs that consume them (3-stage register ring per tile; the sched_barrier keeps the
  // compiler from sinking the reads back next to their use, which would expose one LDS round trip
  // per tap)
  // `tb_c`, `tn_c`: the taps [TB, TB + TN) of the kernel, whose fragments are w[0 .. TN)
  // position descriptor of one tile for this lane: LDS byte address of its sample's channel group gl of
  // block 0, and the byte offsets of the K row / KW column neighbours of its position
  struct TileDesc { const char* base; int roff[K]; int coff[KW]; };
  auto describe = [&](int t, TileDesc& d) {
    const int q = t * 16 + pl;
    const unsigned info = sm.pinfo[q < n_pos ? q : n_pos - 1];
    const int a2 = info & 1023, a1 = (info >> 10) & 1023, sl = info >> 20;
    d.base = (const char*)(in + (size_t)sl * g.CS + gl * g.GS);
    const int* rt = sm.rtab + (dir * g.D1 + a1) * CONV_TAB;
    const int* ct = sm.ctab + (dir * g.D2 + a2) * CONV_TAB;
#pragma unroll
    for (int dd = 0; dd < K; ++dd) d.roff[dd] = rt[dd];
#pragma unroll
    for (int dd = 0; dd < KW; ++dd) d.coff[dd] = ct[dd];
  };
  // the taps [TB, TB + TN) of NTL described tiles against input channel block ci (cib = its byte offset)
  auto multiply = [&](auto nt_c, auto tb_c, auto tn_c, const TileDesc* d, int cib, const f32x4* w, f32x4* acc) {
    constexpr int NTL = decltype(nt_c)::value;
    constexpr int TB = decltype(tb_c)::value, TN = decltype(tn_c)::value;
    f32x4 bq[NTL][3];
#pragma unroll
    for (int h = 0; h < NTL; ++h) {
      bq[h][0] = *(const f32x4*)(d[h].base + cib + d[h].roff[TB / KW] + d[h].coff[TB % KW]);
      if (TN > 1) bq[h][1] = *(const f32x4*)(d[h].base + cib + d[h].roff[TN > 1 ? (TB + 1) / KW : 0] + d[h].coff[TN > 1 ? (TB + 1) % KW : 0]);
    }
#pragma unroll
    for (int tp = 0; tp < TN; ++tp) {
      if (tp + 2 < TN) {
#pragma unroll
        for (int h = 0; h < NTL; ++h)
          bq[h][(tp + 2) % 3] = *(const f32x4*)(d[h].base + cib + d[h].roff[(TB + tp + 2) / KW] + d[h].coff[(TB + tp + 2) % KW]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int h = 0; h < NTL; ++h)
          acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[tp][e], bq[h][tp % 3][e], acc[h], 0, 0, 0);
    }
  };
  // `tb_c`, `tn_c`: the taps [TB, TB + TN) of the kernel, whose fragments are w[0 .. TN)
  auto taps = [&](int t0, auto nt_c, auto tb_c, auto tn_c, int ci, const f32x4* w, f32x4* acc) {
    constexpr int NTL = decltype(nt_c)::value;
    TileDesc d[NTL];
#pragma unroll
    for (int h = 0; h < NTL; ++h) describe(t0 + h, d[h]);
    multiply(nt_c, tb_c, tn_c, d, ci * 4 * g.GS * (int)sizeof(float), w, acc);
  };
  auto store = [&](int t, int co, const f32x4& v) {
    const int q = t * 16 + pl;
    const bool valid = q < n_pos;
    const unsigned info = sm.pinfo[valid ? q : n_pos - 1];
    const int a2 = info & 1023, a1 = (info >> 10) & 1023, sl = info >> 20;
    conv_store_tile(sm, g, out, ep, NCB, co, gl, valid, sl, a1 * g.D2 + a2, v, tape_in, tape_out);
  };
  constexpr std::integral_constant<int, 0> c0{};
  constexpr std::integral_constant<int, 1> c1{};
  constexpr std::integral_constant<int, 2> c2{};

  if constexpr (NCB == 1 && KK <= 49) {
    // every fragment of the layer stays in registers; a wave's tiles in pairs (2w, 2w+1),
    // (2w + 2 NW, ...); an odd last tile runs alone.  With more than 25 taps (6 x 6: 144 weight
    // registers) the pair's second accumulator set would spill, so those kernels take the two tiles
    // one after the other.
    constexpr std::integral_constant<int, KK> ckk{};
    f32x4 w[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) w[t] = *(const f32x4*)(wfrag + ((size_t)t * 64 + lane) * 4);
    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
    if (bias16) bias = *(const f32x4*)(bias16 + 4 * gl);
    constexpr bool PAIR = KK <= 25;
    for (int t0 = 2 * wave; t0 < n_tiles; t0 += 2 * CONV_WAVES) {
      if (PAIR && t0 + 1 < n_tiles) {
        f32x4 acc[2] = {bias, bias};
        taps(t0, c2, c0, ckk, 0, w, acc);
        store(t0, 0, acc[0]); store(t0 + 1, 0, acc[1]);
      } else {
        f32x4 acc[1] = {bias};
        taps(t0, c1, c0, ckk, 0, w, acc);
        store(t0, 0, acc[0]);
        if (!PAIR && t0 + 1 < n_tiles) {
          f32x4 acc1[1] = {bias};
          taps(t0 + 1, c1, c0, ckk, 0, w, acc1);
          store(t0 + 1, 0, acc1[0]);
        }
      }
    }
  } else {
    // NCB x NCB block products.  The wave's tile pairs are taken TC at a time; for one output block
    // their accumulators live in registers while the fragments of (co, ci = 0), (co, 1), ... are
    // loaded in turn and swept over all of them -- in NCH chunks of at most 25 taps (100 registers),
    // so that the accumulators of four tile pairs and a chunk fit the 256 registers of a wave at any
    // kernel size (6 x 6: 2 x 18 taps, 7 x 7: 25 + 24, 8 x 8: 3 x 22, 9 x 9: 4 x 21).  Also the
    // single-block kernels beyond 7 x 7, whose fragments no longer fit the registers at once.
    constexpr int NCH = (KK + 24) / 25, CH = (KK + NCH - 1) / NCH;
    constexpr int TC = 4;
    const int n_pairs = (n_tiles + 1) >> 1;
    // Small kernels with several channel blocks (K + KW <= 6: 3 x 3, 2 x 2, 1 x 1, the chains up to 5 taps):
    // a block pair's products are few (72 MFMAs per tile pair at 3 x 3), and the position descriptors -- a
    // chain of dependent LDS reads: descriptor, wrap tables, then the first operand -- were rebuilt for every
    // one of the NCB x NCB pairs.  They depend on the tile alone: built once per tile group (<= 56 registers)
    // and reused by every pair.
    constexpr bool HOIST = NCB > 1 && K + KW <= 6;
    for (int k0 = 0; wave + CONV_WAVES * k0 < n_pairs; k0 += TC) {
      TileDesc dsc[HOIST ? TC : 1][2];
      if constexpr (HOIST) {
#pragma unroll
        for (int c = 0; c < TC; ++c) {
          const int pi = wave + CONV_WAVES * (k0 + c);
          if (pi < n_pairs) { describe(2 * pi, dsc[c][0]); describe(2 * pi + 1, dsc[c][1]); }   // wave-uniform
        }
      }
#pragma unroll 1
      for (int co = 0; co < NCB; ++co) {
        f32x4 bias = {0.f, 0.f, 0.f, 0.f};
        if (bias16) bias = *(const f32x4*)(bias16 + 16 * co + 4 * gl);
        f32x4 acc[TC][2];
#pragma unroll
        for (int c = 0; c < TC; ++c) { acc[c][0] = bias; acc[c][1] = bias; }
#pragma unroll 1
        for (int ci = 0; ci < NCB; ++ci) {
          const float* wp = wfrag + (size_t)(co * NCB + ci) * KK * 256;
          // (one channel block: the fragments do not depend on the tile loop, and hoisting all of them out
          // of it is what no longer fits the registers -- keep the loads of a chunk next to its use)
          if (NCB == 1) asm volatile("" : "+v"(wp));
          conv_static_for<NCH>([&](auto ch_c) {
            constexpr int TB = decltype(ch_c)::value * CH, TN = TB + CH <= KK ? CH : KK - TB;
            f32x4 w[TN];
#pragma unroll
            for (int t = 0; t < TN; ++t) w[t] = *(const f32x4*)(wp + ((size_t)(TB + t) * 64 + lane) * 4);
#pragma unroll
            for (int c = 0; c < TC; ++c) {
              const int pi = wave + CONV_WAVES * (k0 + c);
              if (pi < n_pairs) {  // wave-uniform
                if constexpr (HOIST)
                  multiply(c2, std::integral_constant<int, TB>{}, std::integral_constant<int, TN>{}, dsc[c],
                           ci * 4 * g.GS * (int)sizeof(float), w, acc[c]);
                else
                  taps(2 * pi, c2, std::integral_constant<int, TB>{}, std::integral_constant<int, TN>{}, ci, w, acc[c]);
              }
            }
          });
        }
#pragma unroll
        for (int c = 0; c < TC; ++c) {
          const int pi = wave + CONV_WAVES * (k0 + c);
          if (pi < n_pairs) {
            store(2 * pi, co, acc[c][0]);
            if (2 * pi + 1 < n_tiles) store(2 * pi + 1, co, acc[c][1]);
          }
        }
      }
    }
  }
}

// Whole forward of the G resident samples: spins (sm.xs) -> sm.red[s] = sum over sites and
// channels of the last feature map (wavefunctions.py:569, 760).  Barriers inside.
template <int K, int KW, int NCB>
__device__ __forceinline__ void conv_forward(const ConvSmem& sm, const ConvGeom& g,
                                             const ConvParams& p, int G, int wave, int lane,
                                             float* tape, long long tape_stride) {
  constexpr size_t WL = (size_t)NCB * NCB * K * KW * 256;     // floats of one layer's fragment image
  constexpr int BL = 16 * NCB;                                 // bias floats per layer
  float* last;
  if (!g.resnet) {
    // [Conv2dPeriodic, nonlinearity] x (n-1), Conv2dPeriodic            (wavefunctions.py:572-575)
    conv_first<K, KW, NCB>(sm, sm.buf0, g, p, G, g.n_conv > 1 ? EP_ACT : EP_LINEAR, wave, lane,
                           (tape && g.n_conv > 1) ? tape : nullptr, 0);
    __syncthreads();
    float* in = sm.buf0; float* out = sm.buf1;
    for (int l = 1; l < g.n_conv; ++l) {
      const bool is_last = l + 1 == g.n_conv;
      conv_layer<K, KW, NCB>(sm, in, out, g, p.wf + (size_t)(l - 1) * WL, p.bias + BL * l, 0, G,
                             is_last ? EP_LINEAR : EP_ACT, wave, lane, nullptr,
                             (tape && !is_last) ? tape + (long long)l * tape_stride : nullptr);
      __syncthreads();
      float* tmp = in; in = out; out = tmp;
    }
    last = in;
  } else {
    // initial_conv, then blocks h <- h + conv2(selu(conv1(h)))          (wavefunctions.py:766-772)
    // tape slot l-1 holds the input of convolution l: h before block k at slot 2k, selu(..) at 2k+1
    conv_first<K, KW, NCB>(sm, sm.buf0, g, p, G, EP_LINEAR, wave, lane, g.n_conv > 1 ? tape : nullptr, 0);
    __syncthreads();
    for (int l = 1; l + 1 < g.n_conv; l += 2) {
      conv_layer<K, KW, NCB>(sm, sm.buf0, sm.buf1, g, p.wf + (size_t)(l - 1) * WL, p.bias + BL * l,
                             0, G, EP_SELU, wave, lane, nullptr,
                             tape ? tape + (long long)l * tape_stride : nullptr);
      __syncthreads();
      conv_layer<K, KW, NCB>(sm, sm.buf1, sm.buf0, g, p.wf + (size_t)l * WL, p.bias + BL * (l + 1),
                             0, G, EP_RESADD, wave, lane, nullptr,
                             (tape && l + 2 < g.n_conv) ? tape + (long long)(l + 1) * tape_stride : nullptr);
      __syncthreads();
    }
    last = sm.buf0;
  }
  // fixed-order reduction: wave w sums samples w, w + CONV_WAVES, ...; padded channels hold exact zeros
  for (int s = wave; s < G; s += CONV_WAVES) {
    const float* m = last + (size_t)s * g.CS;
    float part = 0.f;
    for (int gq = 0; gq < 4 * NCB; ++gq)
      for (int i = lane; i < 4 * g.N; i += 64) part += m[gq * g.GS + i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
    if (lane == 0) sm.red[s] = part;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------- rows
// Amplitudes of a list of rows {chain, bond}: the chain's configuration with the bond's two
// sites exchanged (operators.py:162-163), or the chain itself (bond 0).  Persistent: workgroup b
// takes the row groups b, b + gridDim.x, ... of G rows each.
template <int K, int KW, int NCB>
__global__ __launch_bounds__(CONV_THREADS, CONV_WG_PER_CU) void k_conv_rows(ConvRowsArgs a) {
  extern __shared__ float s_conv[];
  const ConvGeom& g = a.g;
  const int G = a.G;
  const ConvSmem sm = conv_carve(s_conv, g, G);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_rows = a.n_rows_dev ? *a.n_rows_dev : a.n_rows;
  const int xs_stride = conv_xs_stride(g);
  conv_build_pinfo(sm, g, G);
  for (int grp = blockIdx.x; grp * G < n_rows; grp += gridDim.x) {
    // stage the rows' spins, exchange applied
    for (int s = wave; s < G; s += CONV_WAVES) {
      const int row = grp * G + s;
      const bool valid = row < n_rows;
      const int2 ri = a.rowinfo[valid ? row : n_rows - 1];
      const int bs = ri.y;
      const int bond = (bs > 0 ? bs : -bs) - (bs != 0 ? 1 : 0);
      int2 ab = make_int2(0, 0);
      if (bs != 0) ab = a.bonds[bond];
      const float* x = a.configs + (long long)ri.x * g.N;
      const float xi = x[ab.x], xj = x[ab.y];
      for (int i = lane; i < g.N; i += 64) {
        float v = x[i];
        if (bs != 0) v = i == ab.x ? xj : (i == ab.y ? xi : v);
        sm.xs[s * xs_stride + i] = v;
      }
      if (lane == 0) sm.row_chain[s] = valid ? row : -1;
    }
    __syncthreads();
    float* tape = a.tape;
    conv_forward<K, KW, NCB>(sm, g, a.p, G, wave, lane, tape, a.tape_stride);
    for (int s = threadIdx.x; s < G; s += blockDim.x) {
      const int row = grp * G + s;
      if (row < n_rows) {
        const float logit = sm.red[s];
        if (a.ratio) {
          const int2 ri = a.rowinfo[row];
          const int bs = ri.y;
          const int bond = (bs > 0 ? bs : -bs) - (bs != 0 ? 1 : 0);
          a.out[row] = a.half_jx[bond] * vmc_out_ratio(a.oact, logit, a.logit_base[ri.x]);
        } else {
          a.out[row] = logit;
        }
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------- sampler
// n_steps exchange proposals + Metropolis tests per chain (graph_builders.py:38-89) in one launch.
// A workgroup owns G chains; their spins and current logits stay in LDS.  Every proposal is a full
// forward of the proposed configuration (a K x K receptive field grows past the lattice after a few
// layers, so there is no incremental shortcut).
template <int K, int KW, int NCB>
__global__ __launch_bounds__(CONV_THREADS, CONV_WG_PER_CU) void k_conv_sweep(ConvSweepArgs a) {
  extern __shared__ float s_conv[];
  const ConvGeom& g = a.g;
  const int G = a.G;
  const ConvSmem sm = conv_carve(s_conv, g, G);
  float* cur_logit = sm.red + G;               // [G]
  int* prop = (int*)(cur_logit + G);           // [G][2] {i_up, i_dn}
  float* prop_u = (float*)(prop + 2 * G);      // [G]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xs_stride = conv_xs_stride(g);
  const int chain0 = blockIdx.x * G;
  conv_build_pinfo(sm, g, G);
  for (int s = wave; s < G; s += CONV_WAVES) {
    const int c = chain0 + s;
    const bool valid = c < a.B;
    const float* x = a.configs_in + (long long)(valid ? c : a.B - 1) * g.N;
    for (int i = lane; i < g.N; i += 64) sm.xs[s * xs_stride + i] = x[i];
    if (lane == 0) sm.row_chain[s] = valid ? c : -1;
  }
  __syncthreads();
  if (a.cache_in_valid) {
    for (int s = threadIdx.x; s < G; s += blockDim.x) cur_logit[s] = a.logit_in[min(chain0 + s, a.B - 1)];
  } else {
    conv_forward<K, KW, NCB>(sm, g, a.p, G, wave, lane, nullptr, 0);
    for (int s = threadIdx.x; s < G; s += blockDim.x) cur_logit[s] = sm.red[s];
  }
  __syncthreads();
  unsigned long long n_acc = 0;     // thread s counts the accepts of slot s
  const uint2 key = make_uint2(a.seed_lo, a.seed_hi);
  const int nblk = (g.N + 3) >> 2;
  for (long long st = 0; st < a.n_steps || (st == 0 && a.dbg_up); ++st) {
    const unsigned long long step = a.step0 + (unsigned long long)st;
    // proposals: swap_choice = configs * u; lower the up spin with the largest u (argmax), raise
    // the down spin with the largest u (argmin); first index wins ties (graph_builders.py:59-65)
    for (int s = wave; s < G; s += CONV_WAVES) {
      const int c = chain0 + s;
      float* x = sm.xs + s * xs_stride;
      int i_up, i_dn; float u_acc;
      if (a.inj_up) {
        const int cc = min(c, a.B - 1);
        i_up = a.inj_up[cc]; i_dn = a.inj_dn[cc]; u_acc = a.inj_u[cc];
      } else {
        const uint32_t gid = (uint32_t)(a.chain_offset + c);
        float best_hi = -INFINITY, best_lo = INFINITY;
        int idx_hi = 0x7fffffff, idx_lo = 0x7fffffff;
        for (int b = lane; b < nblk; b += 64) {
          const uint4 r = philox4x32_10(make_uint4((uint32_t)b, gid, (uint32_t)step, (uint32_t)(step >> 32)), key);
          const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int i = 4 * b + e;
            if (i < g.N) {
              const float v = x[i] * u32_to_uniform(rr[e]);
              if (v > best_hi) { best_hi = v; idx_hi = i; }
              if (v < best_lo) { best_lo = v; idx_lo = i; }
            }
          }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
          const float oh = __shfl_xor(best_hi, d); const int ih = __shfl_xor(idx_hi, d);
          if (oh > best_hi || (oh == best_hi && ih < idx_hi)) { best_hi = oh; idx_hi = ih; }
          const float ol = __shfl_xor(best_lo, d); const int il = __shfl_xor(idx_lo, d);
          if (ol < best_lo || (ol == best_lo && il < idx_lo)) { best_lo = ol; idx_lo = il; }
        }
        i_up = idx_hi; i_dn = idx_lo;
        const uint4 ra = philox4x32_10(make_uint4(VMC_ACCEPT_BLOCK, gid, (uint32_t)step, (uint32_t)(step >> 32)), key);
        u_acc = u32_to_uniform(ra.x);
      }
      if (a.dbg_up) {
        if (lane == 0 && c < a.B) { a.dbg_up[c] = i_up; a.dbg_dn[c] = i_dn; a.dbg_u[c] = u_acc; }
      } else if (lane == 0) {
        prop[2 * s] = i_up; prop[2 * s + 1] = i_dn; prop_u[s] = u_acc;
        // graph_builders.py:67-71: +2 at the down site, -2 at the up site (scatter_nd sums)
        x[i_dn] += 2.f;
        x[i_up] -= 2.f;
      }
    }
    if (a.dbg_up) break;
    __syncthreads();
    conv_forward<K, KW, NCB>(sm, g, a.p, G, wave, lane, nullptr, 0);
    for (int s = threadIdx.x; s < G; s += blockDim.x) {
      const int c = chain0 + s;
      const float x_new = sm.red[s], x_old = cur_logit[s], u = prop_u[s];
      const bool acc = vmc_out_accept(a.oact, x_new, x_old, u, 0.5f * __logf(u));
      float* x = sm.xs + s * xs_stride;
      if (acc) {
        cur_logit[s] = x_new;
      } else {
        x[prop[2 * s + 1]] -= 2.f;
        x[prop[2 * s]] += 2.f;
      }
      if (c < a.B) {
        n_acc += acc ? 1ull : 0ull;
        if (a.acc_mask) a.acc_mask[c] = acc ? 1 : 0;
      }
    }
    __syncthreads();
  }
  if (a.dbg_up) return;
  for (int s = wave; s < G; s += CONV_WAVES) {
    const int c = chain0 + s;
    if (c < a.B) {
      float* dst = a.configs + (long long)c * g.N;
      for (int i = lane; i < g.N; i += 64) dst[i] = sm.xs[s * xs_stride + i];
      if (lane == 0) a.logit[c] = cur_logit[s];
    }
  }
  if (n_acc) atomicAdd(a.accepted, n_acc);
}

// ---------------------------------------------------------------------------------- backward
// d logit / d (output of convolution l) for every l, from the forward tapes.  Conv2DNetwork:
// delta_{n-1} = oscale, delta_{l-1} = f'(a_l) (.) convT_l(delta_l).  ResNet2D: D = oscale;
// per block (last first) delta_{2k+2} = D, delta_{2k+1} = selu'(t_k) (.) convT_{2k+2}(D),
// D += convT_{2k+1}(delta_{2k+1}); delta_0 = D.  convT is the same tile loop with the flipped,
// transposed fragment image and the padding roles exchanged.
template <int K, int KW, int NCB>
__global__ __launch_bounds__(CONV_THREADS, CONV_WG_PER_CU) void k_conv_back(ConvBackArgs a) {
  constexpr size_t WL = (size_t)NCB * NCB * K * KW * 256;
  extern __shared__ float s_conv[];
  const ConvGeom& g = a.g;
  const int G = a.G;
  const ConvSmem sm = conv_carve(s_conv, g, G);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  conv_build_pinfo(sm, g, G);
  const int n = g.n_conv;
  for (int grp = blockIdx.x; grp * G < a.B; grp += gridDim.x) {
    for (int s = threadIdx.x; s < G; s += blockDim.x) sm.row_chain[s] = grp * G + s < a.B ? grp * G + s : -1;
    __syncthreads();
    // seed: d logit / d (last feature map) = oscale on the real channels
    for (int s = wave; s < G; s += CONV_WAVES) {
      const int row = sm.row_chain[s];
      const float sc = row >= 0 ? a.oscale[row] : 0.f;
      float* d = sm.buf0 + (size_t)s * g.CS;
      for (int gq = 0; gq < 4 * NCB; ++gq)
        for (int i = lane; i < 4 * g.N; i += 64) {
          const float v = (4 * gq + (i & 3)) < g.F ? sc : 0.f;
          d[gq * g.GS + i] = v;
          if (row >= 0) a.delta[(long long)(n - 1) * a.delta_stride + ((long long)row * 4 * NCB + gq) * g.GS + i] = v;
        }
    }
    __syncthreads();
    if (!g.resnet) {
      float* in = sm.buf0; float* out = sm.buf1;
      for (int l = n - 1; l >= 1; --l) {
        conv_layer<K, KW, NCB>(sm, in, out, g, a.p.wb + (size_t)(l - 1) * WL, nullptr, 1, G,
                               EP_BACK_DACT, wave, lane, a.tape + (long long)(l - 1) * a.tape_stride,
                               a.delta + (long long)(l - 1) * a.delta_stride);
        __syncthreads();
        float* tmp = in; in = out; out = tmp;
      }
    } else {
      for (int l = n - 1; l >= 2; l -= 2) {   // block with convolutions l-1 (first) and l (second)
        if (l != n - 1) {                     // delta_l = D (the seed above covers the last block)
          for (int s = wave; s < G; s += CONV_WAVES) {
            const int row = sm.row_chain[s];
            const float* d = sm.buf0 + (size_t)s * g.CS;
            if (row >= 0)
              for (int gq = 0; gq < 4 * NCB; ++gq)
                for (int i = lane; i < 4 * g.N; i += 64)
                  a.delta[(long long)l * a.delta_stride + ((long long)row * 4 * NCB + gq) * g.GS + i] = d[gq * g.GS + i];
          }
        }
        conv_layer<K, KW, NCB>(sm, sm.buf0, sm.buf1, g, a.p.wb + (size_t)(l - 1) * WL, nullptr, 1, G,
                               EP_BACK_SELU, wave, lane, a.tape + (long long)(l - 1) * a.tape_stride,
                               a.delta + (long long)(l - 1) * a.delta_stride);
        __syncthreads();
        conv_layer<K, KW, NCB>(sm, sm.buf1, sm.buf0, g, a.p.wb + (size_t)(l - 2) * WL, nullptr, 1, G,
                               EP_BACK_ADD, wave, lane, nullptr, l == 2 ? a.delta : nullptr);
        __syncthreads();
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------- SR row dot
// Stochastic reconfiguration (extension): t_b = O_b . p = sum_l <delta_l[b], conv(in_l[b], V_l) + v_l>
// over the stored samples, with (V_l, v_l) the slice of the CG direction p for convolution l (packed
// like a parameter set) and in_l / delta_l the taped inputs and back-propagated d logit / d (output)
// of the gradient path.  Per layer the taped input is staged in LDS, the convolution runs with the
// EP_DOT epilogue (accumulate (conv + bias) * delta per element) and the per-sample sum is the same
// fixed-order reduction as the forward's.  Persistent over groups of G stored samples.
template <int K, int KW, int NCB>
__global__ __launch_bounds__(CONV_THREADS, CONV_WG_PER_CU) void k_conv_sr_rowdot(ConvSrRowdotArgs a) {
  constexpr size_t WL = (size_t)NCB * NCB * K * KW * 256;
  constexpr int BL = 16 * NCB;
  extern __shared__ float s_conv[];
  const ConvGeom& g = a.g;
  const int G = a.G;
  const ConvSmem sm = conv_carve(s_conv, g, G);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xs_stride = conv_xs_stride(g);
  const bool tape_is_z = !g.resnet && g.hact == VMC_ACT_COS_;
  conv_build_pinfo(sm, g, G);
  for (int grp = blockIdx.x; grp * G < a.n_rows; grp += gridDim.x) {
    for (int s = wave; s < G; s += CONV_WAVES) {
      const int row = grp * G + s;
      const bool valid = row < a.n_rows;
      const float* x = a.configs + (long long)(valid ? row : a.n_rows - 1) * g.N;
      for (int i = lane; i < g.N; i += 64) sm.xs[s * xs_stride + i] = x[i];
      if (lane == 0) sm.row_chain[s] = valid ? row : -1;
    }
    for (int i = threadIdx.x; i < G * g.CS; i += blockDim.x) sm.buf1[i] = 0.f;     // the accumulation buffer
    __syncthreads();
    conv_first<K, KW, NCB>(sm, sm.buf1, g, a.p, G, EP_DOT, wave, lane, const_cast<float*>(a.delta), 0);
    __syncthreads();
    for (int l = 1; l < g.n_conv; ++l) {
      // stage the taped input of convolution l (slot l - 1): [row][CS] -> buf0[s][CS]
      const float* tp = a.tape + (long long)(l - 1) * a.tape_stride;
      for (int s = 0; s < G; ++s) {
        const int row = sm.row_chain[s];
        const float* src = tp + (long long)(row >= 0 ? row : 0) * g.CS;
        for (int i = 4 * threadIdx.x; i < g.CS; i += 4 * blockDim.x) {
          f32x4 v = *(const f32x4*)(src + i);
          if (tape_is_z) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = vmc_act_rt(VMC_ACT_COS_, v[r]);
          }
          *(f32x4*)(sm.buf0 + (size_t)s * g.CS + i) = v;
        }
      }
      __syncthreads();
      conv_layer<K, KW, NCB>(sm, sm.buf0, sm.buf1, g, a.p.wf + (size_t)(l - 1) * WL, a.p.bias + BL * l, 0, G,
                             EP_DOT, wave, lane, a.delta + (long long)l * a.delta_stride, nullptr);
      __syncthreads();
    }
    for (int s = wave; s < G; s += CONV_WAVES) {
      const float* m = sm.buf1 + (size_t)s * g.CS;
      float part = 0.f;
      for (int gq = 0; gq < 4 * NCB; ++gq)
        for (int i = lane; i < 4 * g.N; i += 64) part += m[gq * g.GS + i];
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
      const int row = sm.row_chain[s];
      if (lane == 0 && row >= 0) a.t[row] = part;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------- weight gradient
// sum_b (1 | w_b) * d logit_b / d W_l for every convolution: dW[tap][cin][cout] =
// sum_{b, pos} in_l[b, pos + tap, cin] * delta_l[b, pos, cout]  (and the bias: sum of delta_l).
// Grid (slice, layer): a workgroup walks the samples slice, slice + n_slices, ... (so that at any
// moment the workgroups read neighbouring samples).  A sample is staged in LDS in ONE padded site
// numbering q = a1 (D2 + KW - 1) + a2: the convolution's input with its periodic halo,
// [D1 + K - 1][D2 + KW - 1][16 NCB channels], and delta [D1][D2 + KW - 1][16 NCB] with zeros in the
// halo columns -- so position q under tap (t1, t2) reads input site q + t1 (D2 + KW - 1) + t2: the
// walk over positions is a pointer increment, a tap a constant offset, and the halo positions add
// zeros (1.4 x the products at 10 x 10, k = 5, in exchange for no address arithmetic: the loop was
// VALU-issue bound).  Wave w owns the items w, w + 8, ... (the taps, then the bias); the reduction
// over positions is the k index of the MFMA (4 positions per instruction): A = input (lane = cin;
// the bias item: ones), B = delta (lane = cout), and a second accumulator takes w_b * delta; NCB x NCB
// channel-block products per tap.  Partial sums go to ws[slice][layer]; k_conv_dw_reduce adds the
// slices in a fixed order into the accumulators.  (cos: the tape holds z, the input is cos z.)
// Latency rules: (1) the next sample is fetched into registers while the current one is multiplied,
// with no branch around a load (the compiler drains the queue at every join); (2) the LDS operands
// of position quad q + 4 are read before the products of quad q are issued; (3) BOTH (both sums, or
// the weighted one only: the SR matvec) and FIRST (the one-channel first layer, whose taps are the
// MFMA's m index) are compile-time, so that no variant pays for another's registers.
#define DW_WAVES 8   // the weight-gradient kernel splits the items over 8 waves
template <int K, int KW, int NCB, bool BOTH, bool FIRST>
__device__ __forceinline__ void conv_dw_body(const ConvDwArgs& a, float* s_dw) {
  constexpr int NCO = plan_conv_dw_nco(K, KW, NCB);              // output channel blocks of this workgroup
  constexpr int NTP = plan_conv_dw_parts(K, KW, NCB);            // parts the items of a layer > 0 are cut into (grid z)
  constexpr int KK = K * KW;
  constexpr int NI = KK + 1;                                     // items of a layer > 0: the taps, then the bias
  constexpr int T0 = (KK + 15) / 16;                             // tap tiles of the first layer
  constexpr int TPW = FIRST ? (T0 + DW_WAVES - 1) / DW_WAVES : (NI + DW_WAVES * NTP - 1) / (DW_WAVES * NTP);   // items per wave
  static_assert(DW_WAVES == PLAN_DW_WAVES, "plan.hpp sizes the item parts for this many waves");
  constexpr int CW = 16 * NCB;                                   // staged input channels per site
  constexpr int CWD = 16 * NCO;                                  // staged delta channels per site (this workgroup's output blocks)
  constexpr int CI = FIRST ? 1 : NCB;                            // input channel blocks
  constexpr int NA = TPW * CI * NCO;
  constexpr int GQ = 4 * NCB, GQD = 4 * NCO;                     // f32x4 channel groups per site: input, delta
  constexpr int U = NCB == 1 ? 4 : (NCB == 2 ? 5 : 7);           // prefetched vectors per thread
  constexpr int WGT = DW_WAVES * 64;
  const ConvGeom& g = a.g;
  const int l = blockIdx.y;
  const int co0 = (blockIdx.z % (NCB / NCO)) * NCO;   // first output channel block of this workgroup
  const int part = blockIdx.z / (NCB / NCO);       // its part of the items (uniform)
  if (FIRST && part > 0) return;                   // the first layer's tap tiles fit one part
  const int i0 = part * TPW;                       // first item slot of this part
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ml = lane & 15, gl = lane >> 4;
  const int D2p = g.D2 + KW - 1, NPAD = (g.D1 + K - 1) * D2p;
  // a sample is taken in bands of RB lattice rows (all of them when that fits LDS: a.band_rows)
  const int RB = a.band_rows;
  const int NQ = (RB * D2p + 3) & ~3;              // positions walked per band (padded numbering), whole quads
  const int NIN = NQ + (K - 1) * D2p + KW;         // input sites a product can touch
  // LDS: delta [NQ][CWD]; input [NIN][CW] (first layer: spins, one float per site); the halo map
  // (source site of every padded input site), the position map (padded number of every site), ones
  float* s_dl = s_dw;
  float* s_in = s_dw + (size_t)NQ * CWD;
  int* s_map = (int*)(s_in + (size_t)NIN * (FIRST ? 1 : CW));
  int* s_pos = s_map + NPAD;
  float* s_one = (float*)(s_pos + g.N);            // [CW]: 1 for channel block 0 (the bias item's A operand)
  const int b0 = blockIdx.x, b1 = a.B, bstep = a.n_slices;
  const bool tape_is_z = !g.resnet && g.hact == VMC_ACT_COS_;
  f32x4 acc1[BOTH ? NA : 1], acc2[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) { acc2[i] = f32x4{0.f, 0.f, 0.f, 0.f}; if (BOTH) acc1[i] = acc2[i]; }
  f32x4 bacc1[NCO], bacc2[NCO];                    // first layer's bias: A = ones (wave DW_WAVES - 1)
#pragma unroll
  for (int i = 0; i < NCO; ++i) { bacc1[i] = f32x4{0.f, 0.f, 0.f, 0.f}; bacc2[i] = bacc1[i]; }
  // once: zeros (delta's halo columns and tail, the input's tail stay zero for every sample), maps
  for (int i = threadIdx.x; i < NQ * CWD + NIN * (FIRST ? 1 : CW); i += WGT) s_dw[i] = 0.f;
  if (threadIdx.x < CW) s_one[threadIdx.x] = threadIdx.x < 16 ? 1.f : 0.f;
  for (int i = threadIdx.x; i < NPAD; i += WGT) {
    const int p1 = i / D2p, p2 = i - p1 * D2p;
    s_map[i] = wrap(p1 - g.lo, g.D1) * g.D2 + wrap(p2 - g.lo2, g.D2);
  }
  for (int i = threadIdx.x; i < g.N; i += WGT) {
    const int a1 = i / g.D2;
    s_pos[i] = a1 * D2p + (i - a1 * g.D2);
  }
  __syncthreads();
  // one flattened item list per (sample, band): delta rows [r0, r0 + rows), then (layers > 0) the
  // input's padded rows [r0, r0 + rows + K - 1)
  const int in_off = NQ * CWD;                     // s_in relative to s_dw, in floats
  auto band_rows_of = [&](int r0) { return min(RB, g.D1 - r0); };
  auto band_nd = [&](int rows) { return GQD * rows * g.D2; };
  auto band_ntot = [&](int rows) { return GQD * rows * g.D2 + (FIRST ? 0 : GQ * (rows + K - 1) * D2p); };
  auto item = [&](int i, int r0, int nd, int& soff, int& dst) {
    const bool isd = i < nd;
    const int k = isd ? i : i - nd;
    // ps: site within the band / padded site within the band; gq: channel group (constant divisors)
    const int ps = isd ? k / GQD : k / GQ, gq = isd ? k - ps * GQD : k - ps * GQ;
    const int site = isd ? r0 * g.D2 + ps : s_map[r0 * D2p + ps];
    const int q = isd ? s_pos[site] - r0 * D2p : ps;
    soff = ((isd ? 4 * co0 : 0) + gq) * g.GS + 4 * site;
    dst = isd ? q * CWD + 4 * gq : in_off + q * CW + 4 * gq;
  };
  auto put = [&](int dst, f32x4 v) {
    if (tape_is_z && dst >= in_off) {
      const int gq = ((dst - in_off) >> 2) % GQ;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (4 * gq + r) < g.F ? vmc_act_rt(VMC_ACT_COS_, v[r]) : 0.f;
    }
    *(f32x4*)(s_dw + dst) = v;
  };
  // the first U x 512 items of the next sample travel while the current one is multiplied (a 10 x 10
  // lattice with k = 5 has 1184 items at 16 filters, 2368 at 32); larger lattices stage the rest late
  f32x4 pre[U];
  float wb_next = 0.f, spin_next = 0.f;
  auto prefetch = [&](int b, int r0) {
    const float* dsrc = a.delta + (long long)l * a.delta_stride + (long long)b * g.CS;
    const float* isrc = a.tape + (long long)(FIRST ? 0 : l - 1) * a.tape_stride + (long long)b * g.CS;
    const int rows = band_rows_of(r0), nd = band_nd(rows), ntot = band_ntot(rows);
    wb_next = a.w[b];
    if (FIRST)
      spin_next = a.configs[(long long)b * g.N + s_map[r0 * D2p + min((int)threadIdx.x, (rows + K - 1) * D2p - 1)]];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      // items past the end repeat the last one and are never stored
      const int i = min((int)threadIdx.x + j * WGT, ntot - 1);
      int soff, dst;
      item(i, r0, nd, soff, dst);
      pre[j] = *(const f32x4*)((dst < in_off ? dsrc : isrc) + soff);
    }
  };
  // per item: A operand pointer of this lane at position quad 0, and its advance per quad.  Layers
  // > 0: lane (ml, gl) = (cin, position), item = tap or bias (the ones row, which does not advance);
  // an empty slot reads tap 0 and is not multiplied.  First layer: lane ml = tap 16 tt + ml
  int ap0[TPW], astep[TPW];                        // offsets into s_dw, in floats (LDS address space kept)
  const int one_off = (int)(s_one - s_dw), sin_off = (int)(s_in - s_dw);
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    if (FIRST) {
      int tap = 16 * (wave + i * DW_WAVES) + ml;
      tap = tap < KK ? tap : 0;
      ap0[i] = sin_off + gl + (tap / KW) * D2p + tap % KW;
      astep[i] = 4;
    } else {
      const int it = wave + (i0 + i) * DW_WAVES;   // wave-uniform
      const int tp = it < KK ? it : 0;
      ap0[i] = it == KK ? one_off + ml : sin_off + ml + (gl + (tp / KW) * D2p + tp % KW) * CW;
      astep[i] = it == KK ? 0 : 4 * CW;
    }
  }
  if (b0 < b1) prefetch(b0, 0);
  for (int b = b0; b < b1; b += bstep)
  for (int r0 = 0; r0 < g.D1; r0 += RB) {
    const int rows = band_rows_of(r0), nd = band_nd(rows), ntot = band_ntot(rows);
    const int nq = (rows * D2p + 3) & ~3;          // positions walked in this band
    __syncthreads();
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int i = threadIdx.x + j * WGT;
      if (i < ntot) {
        int soff, dst;
        item(i, r0, nd, soff, dst);
        put(dst, pre[j]);
      }
    }
    if (ntot > U * WGT) {                          // larger lattices: the remainder, not prefetched
      const float* dsrc = a.delta + (long long)l * a.delta_stride + (long long)b * g.CS;
      const float* isrc = a.tape + (long long)(FIRST ? 0 : l - 1) * a.tape_stride + (long long)b * g.CS;
      for (int i = threadIdx.x + U * WGT; i < ntot; i += WGT) {
        int soff, dst;
        item(i, r0, nd, soff, dst);
        put(dst, *(const f32x4*)((dst < in_off ? dsrc : isrc) + soff));
      }
    }
    if (FIRST) {
      const int nsp = (rows + K - 1) * D2p;
      if ((int)threadIdx.x < nsp) s_in[threadIdx.x] = spin_next;
      for (int i = threadIdx.x + WGT; i < nsp; i += WGT) s_in[i] = a.configs[(long long)b * g.N + s_map[r0 * D2p + i]];
    }
    // a short last band: the walk's last quad may reach into rows an earlier band wrote
    if (rows < RB && (int)threadIdx.x < (nq - rows * D2p) * CWD) s_dl[rows * D2p * CWD + threadIdx.x] = 0.f;
    const float wb = wb_next;
    __syncthreads();
    {
      const bool more_bands = r0 + RB < g.D1;
      const int nb = more_bands ? b : b + bstep, nr0 = more_bands ? r0 + RB : 0;
      if (nb < b1) prefetch(nb, nr0);
    }
    // the walk over position quads, operands read one quad ahead
    int dp = gl * CWD + ml;
    int ap[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) ap[i] = ap0[i];
    float dv[NCO], av[TPW][CI], dvn[NCO], avn[TPW][CI];
#pragma unroll
    for (int co = 0; co < NCO; ++co) dv[co] = s_dw[dp + 16 * co];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
      for (int ci = 0; ci < CI; ++ci) av[i][ci] = s_dw[ap[i] + 16 * ci];
    for (int q = 0; q < nq; q += 4) {
      dp += 4 * CWD;                               // (the quad past the end reads the input region: dropped)
#pragma unroll
      for (int co = 0; co < NCO; ++co) dvn[co] = s_dw[dp + 16 * co];
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        ap[i] += astep[i];
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) avn[i][ci] = s_dw[ap[i] + 16 * ci];
      }
      float dv2[NCO];
#pragma unroll
      for (int co = 0; co < NCO; ++co) dv2[co] = dv[co] * wb;
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const bool full = FIRST ? (i * DW_WAVES + DW_WAVES - 1 < T0) : (NTP == 1 && i * DW_WAVES + DW_WAVES - 1 < NI);
        if (full || wave + (i0 + i) * DW_WAVES < (FIRST ? T0 : NI)) {   // last slot: partly filled (wave-uniform)
#pragma unroll
          for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int co = 0; co < NCO; ++co) {
              const int ai = (i * CI + ci) * NCO + co;
              if (BOTH) acc1[ai] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][ci], dv[co], acc1[ai], 0, 0, 0);
              acc2[ai] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][ci], dv2[co], acc2[ai], 0, 0, 0);
            }
        }
      }
      if (FIRST && wave == DW_WAVES - 1) {
#pragma unroll
        for (int co = 0; co < NCO; ++co) {
          if (BOTH) bacc1[co] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, dv[co], bacc1[co], 0, 0, 0);
          bacc2[co] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, dv2[co], bacc2[co], 0, 0, 0);
        }
      }
#pragma unroll
      for (int co = 0; co < NCO; ++co) dv[co] = dvn[co];
#pragma unroll
      for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) av[i][ci] = avn[i][ci];
    }
  }
  // partial sums: ws[slice][layer][2][(KK*CW + 1) * CW]: row (tap * CW + cin) or KK*CW = bias, col cout
  // (the weighted-only variant leaves the first half untouched: the reduce skips it as well)
  const size_t rows = (size_t)KK * CW + 1;
  float* w1 = a.ws + (((size_t)blockIdx.x * g.n_conv + l) * 2) * rows * CW;
  float* w2 = w1 + rows * CW;
  if (FIRST) {
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int tt = wave + i * DW_WAVES;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tap = 16 * tt + 4 * gl + r;          // accumulator row = tap
        if (tt < T0 && tap < KK) {
#pragma unroll
          for (int co = 0; co < NCO; ++co) {
            if (BOTH) w1[(size_t)tap * CW * CW + 16 * (co0 + co) + ml] = acc1[i * NCO + co][r];   // cin 0
            w2[(size_t)tap * CW * CW + 16 * (co0 + co) + ml] = acc2[i * NCO + co][r];
          }
        }
      }
    }
    if (wave == DW_WAVES - 1 && gl == 0) {
#pragma unroll
      for (int co = 0; co < NCO; ++co) {
        if (BOTH) w1[(size_t)KK * CW * CW + 16 * (co0 + co) + ml] = bacc1[co][0];
        w2[(size_t)KK * CW * CW + 16 * (co0 + co) + ml] = bacc2[co][0];
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int it = wave + (i0 + i) * DW_WAVES;
      if (it < KK) {
#pragma unroll
        for (int ci = 0; ci < NCB; ++ci)
#pragma unroll
          for (int co = 0; co < NCO; ++co)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int ai = (i * NCB + ci) * NCO + co;
              const size_t o = ((size_t)it * CW + 16 * ci + 4 * gl + r) * CW + 16 * (co0 + co) + ml;   // row cin = 16 ci + 4g + r
              if (BOTH) w1[o] = acc1[ai][r];
              w2[o] = acc2[ai][r];
            }
      } else if (it == KK && gl == 0) {                // bias: every row of the ones product is the sum
#pragma unroll
        for (int co = 0; co < NCO; ++co) {
          if (BOTH) w1[(size_t)KK * CW * CW + 16 * (co0 + co) + ml] = acc1[(i * NCB) * NCO + co][0];
          w2[(size_t)KK * CW * CW + 16 * (co0 + co) + ml] = acc2[(i * NCB) * NCO + co][0];
        }
      }
    }
  }
}

// Grid (slice, layer, z): z = (output channel block group, item part).  Up to two channel blocks and
// 7 x 7 taps a workgroup takes every output block and every tap of its layer; beyond, the NCB x NCB
// accumulator pairs per tap would not fit the register file, so a workgroup stages the whole input
// but only NCO blocks of delta, and takes one of NTP parts of the taps (plan_conv_dw_nco / _parts)
template <int K, int KW, int NCB, bool BOTH>
__global__ __launch_bounds__(DW_WAVES * 64, (NCB == 1 && K * KW <= 49) ? 4 : 2) void k_conv_dw(ConvDwArgs a) {   // 16 filters: two workgroups per CU
  extern __shared__ float s_dw[];
  if (blockIdx.y == 0) conv_dw_body<K, KW, NCB, BOTH, true>(a, s_dw);
  else conv_dw_body<K, KW, NCB, BOTH, false>(a, s_dw);
}

template <typename Kern, typename Args>
hipError_t launch_k(Kern kern, dim3 grid, size_t lds, hipStream_t s, const Args& a, int threads = CONV_THREADS) {
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, grid, dim3(threads), lds, s, a);
  return hipGetLastError();
}

// (kernel_size, taps along axis 2): square kernels (Conv2dPeriodic) and k x 1 (Conv1dPeriodic)
#define CONV_DISPATCH_K(G_, CALL)                                                       \
  switch ((G_).KW == 1 ? -(G_).K : (G_).K) {                                            \
    case 1: case -1: { constexpr int KK_ = 1, KW_ = 1; CALL; } break;                    \
    case 2: { constexpr int KK_ = 2, KW_ = 2; CALL; } break;                            \
    case 3: { constexpr int KK_ = 3, KW_ = 3; CALL; } break;                            \
    case 4: { constexpr int KK_ = 4, KW_ = 4; CALL; } break;                            \
    case 5: { constexpr int KK_ = 5, KW_ = 5; CALL; } break;                            \
    case 6: { constexpr int KK_ = 6, KW_ = 6; CALL; } break;                            \
    case 7: { constexpr int KK_ = 7, KW_ = 7; CALL; } break;                            \
    case 8: { constexpr int KK_ = 8, KW_ = 8; CALL; } break;                            \
    case 9: { constexpr int KK_ = 9, KW_ = 9; CALL; } break;                            \
    case -2: { constexpr int KK_ = 2, KW_ = 1; CALL; } break;                           \
    case -3: { constexpr int KK_ = 3, KW_ = 1; CALL; } break;                           \
    case -4: { constexpr int KK_ = 4, KW_ = 1; CALL; } break;                           \
    case -5: { constexpr int KK_ = 5, KW_ = 1; CALL; } break;                           \
    case -6: { constexpr int KK_ = 6, KW_ = 1; CALL; } break;                           \
    case -7: { constexpr int KK_ = 7, KW_ = 1; CALL; } break;                           \
    case -8: { constexpr int KK_ = 8, KW_ = 1; CALL; } break;                           \
    case -9: { constexpr int KK_ = 9, KW_ = 1; CALL; } break;                           \
    default: return hipErrorInvalidValue;                                               \
  }

// the five launchers for one NCB (conv.hip: 1; conv_wide.hpp: 2, 3, 4)
template <int NCB>
hipError_t conv_launch_rows_t(hipStream_t s, const ConvRowsArgs& a, dim3 grid, size_t lds) {
  CONV_DISPATCH_K(a.g, return launch_k(k_conv_rows<KK_, KW_, NCB>, grid, lds, s, a));
  return hipSuccess;
}
template <int NCB>
hipError_t conv_launch_sweep_t(hipStream_t s, const ConvSweepArgs& a, dim3 grid, size_t lds) {
  CONV_DISPATCH_K(a.g, return launch_k(k_conv_sweep<KK_, KW_, NCB>, grid, lds, s, a));
  return hipSuccess;
}
template <int NCB>
hipError_t conv_launch_back_t(hipStream_t s, const ConvBackArgs& a, dim3 grid, size_t lds) {
  CONV_DISPATCH_K(a.g, return launch_k(k_conv_back<KK_, KW_, NCB>, grid, lds, s, a));
  return hipSuccess;
}
template <int NCB>
hipError_t conv_launch_sr_rowdot_t(hipStream_t s, const ConvSrRowdotArgs& a, dim3 grid, size_t lds) {
  CONV_DISPATCH_K(a.g, return launch_k(k_conv_sr_rowdot<KK_, KW_, NCB>, grid, lds, s, a));
  return hipSuccess;
}
template <int NCB>
hipError_t conv_launch_dw_t(hipStream_t s, const ConvDwArgs& a, dim3 grid, size_t lds) {
  if (a.g1) { CONV_DISPATCH_K(a.g, return launch_k(k_conv_dw<KK_, KW_, NCB, true>, grid, lds, s, a, DW_WAVES * 64)); }
  else { CONV_DISPATCH_K(a.g, return launch_k(k_conv_dw<KK_, KW_, NCB, false>, grid, lds, s, a, DW_WAVES * 64)); }
  return hipSuccess;
}

}  // namespace
