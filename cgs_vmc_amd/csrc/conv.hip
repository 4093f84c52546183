// Convolutional ansatz kernels (Conv2DNetwork / ResNet2D, wavefunctions.py:531-615, 710-809, and
// their 1-D siblings Conv1DNetwork / ResNet1D, 455-527, 618-707, as k x 1 taps on an N x 1 lattice)
// for gfx950.  See DESIGN.md 4 "Convolutional ansatz types".
//
// A periodic convolution with <= 16 channels is an implicit GEMM whose output tile is exactly one
// v_mfma_f32_16x16x4_f32 tile: 16 output channels x 16 lattice positions, reduced over
// (tap, input channel) four input channels at a time.  The A operand is a weight fragment (all
// K*K*4 of them stay in registers for the whole layer), the B operand is one ds_read_b128 of the
// input feature map per tap: lane (p, g) reads channels 4g..4g+3 of the site `tap` away from
// position p.  The accumulator comes out with the position on the lane and channel 4g+r on
// register r, which is the layout the next layer reads, so the epilogue is one ds_write_b128.
// Feature maps of the G samples a workgroup has in flight never leave LDS between layers.
#include "conv.hpp"
#include <cstdlib>
#include <type_traits>

// 4 waves per workgroup (one per SIMD) and two workgroups per CU: the two co-resident workgroups are
// never in step, so the serial phases of one (row staging behind dependent global loads, the
// per-layer weight-fragment reload, barriers, the final reduction) run under the MFMAs of the
// other (one 4-wave workgroup per CU: 0.58 of the fp32-MFMA peak against 0.72 for two).
#define CONV_WAVES 4
#define CONV_THREADS (CONV_WAVES * 64)
#define CONV_LDS_PER_WG (80 * 1024)   // two workgroups share the 160 KiB of a CU
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

// f'(z) from a = f(z) (run-time activation id; the cosine is not offered by the convolutional
// ansatz kernels: its derivative needs z itself)
__device__ __forceinline__ float dact_from_a_rt(int act, float a) {
  switch (act) {
    case VMC_ACT_RELU_: return a > 0.f ? 1.f : 0.f;
    case VMC_ACT_EXP_: return a;
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
       EP_BACK_ADD = 6 }; // out = out + acc

// per-position descriptor: sample slot, lattice coordinates (built once per kernel, LDS)
__device__ __forceinline__ unsigned pack_pos(int s, int a1, int a2) {
  return (unsigned)a2 | ((unsigned)a1 << 10) | ((unsigned)s << 20);
}

struct ConvSmem {
  float* xs;        // [G][XS] spins
  float* buf0;      // [G][4 GS]
  float* buf1;      // [G][4 GS]
  unsigned* pinfo;  // [G N] pack_pos
  int* row_chain;   // [G] chain (or row) of each slot, -1 = empty
  float* red;       // [G] reduced logits
  // periodic neighbour tables, built once per kernel (the wrap arithmetic costs ~15 VALU
  // instructions per tap column / row and tile otherwise): rtab[dir][a1][d] = 16 D2 ((a1 + d - lo)
  // mod D1), ctab[dir][a2][d] = 16 ((a2 + d - lo) mod D2) in bytes (one v_add3 per tap and tile), lo = g.lo (dir 0: forward) or
  // g.hi (dir 1: transposed convolution); rows of 8 ints
  int* rtab;        // [2][D1][8]
  int* ctab;        // [2][D2][8]
};

__device__ __forceinline__ int conv_xs_stride(const ConvGeom& g) { return (g.N + 3) & ~3; }

__device__ __forceinline__ ConvSmem conv_carve(float* base, const ConvGeom& g, int G) {
  ConvSmem s;
  s.buf0 = base;
  s.buf1 = s.buf0 + (size_t)G * 4 * g.GS;
  s.xs = s.buf1 + (size_t)G * 4 * g.GS;
  s.pinfo = (unsigned*)(s.xs + (size_t)G * conv_xs_stride(g));
  s.row_chain = (int*)(s.pinfo + (size_t)G * g.N);
  s.red = (float*)(s.row_chain + G);
  s.rtab = (int*)(s.red + 6 * G);       // red, cur_logit, prop[2], prop_u of the sampler + spare
  s.ctab = s.rtab + 2 * g.D1 * 8;
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
  for (int i = threadIdx.x; i < 2 * (g.D1 + g.D2) * 8; i += blockDim.x) {
    const bool is_r = i < 2 * g.D1 * 8;
    const int j = is_r ? i : i - 2 * g.D1 * 8, D = is_r ? g.D1 : g.D2;
    const int dir = j / (D * 8), a = (j / 8) % D, d = j & 7;
    const int lo = is_r ? (dir ? g.hi : g.lo) : (dir ? g.hi2 : g.lo2);
    const int w = ((a + min(d, (is_r ? g.K : g.KW) - 1) - lo) % D + D) % D;
    (is_r ? sm.rtab : sm.ctab)[j] = is_r ? 16 * g.D2 * w : 16 * w;     // byte offsets
  }
}

// First convolution (one input channel, layers.py:151-160 on the reshaped spins): the k index of
// the MFMA runs over the taps, four per instruction.
template <int K, int KW>
__device__ __forceinline__ void conv_first(const ConvSmem& sm, float* out, const ConvGeom& g,
                                           const ConvParams& p, int G, int ep, int wave, int lane,
                                           float* tape_out, long long tape_rows) {
  constexpr int Q0 = (K * KW + 3) / 4;
  const int pl = lane & 15, gl = lane >> 4;
  float w0[Q0];
#pragma unroll
  for (int q = 0; q < Q0; ++q) w0[q] = p.w0[q * 64 + lane];
  const f32x4 bias = *(const f32x4*)(p.bias + 4 * gl);
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
    f32x4 acc = bias;
    float bx[Q0];
#pragma unroll
    for (int qq = 0; qq < Q0; ++qq) bx[qq] = xs[(sm.rtab[a1 * 8 + d1[qq]] + sm.ctab[a2 * 8 + d2[qq]]) >> 4];
#pragma unroll
    for (int qq = 0; qq < Q0; ++qq) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[qq], bx[qq], acc, 0, 0, 0);
    if (ep == EP_ACT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = vmc_act_rt(g.hact, acc[r]);
    }
    if (valid) {
      const int site = a1 * g.D2 + a2;
      *(f32x4*)(out + (size_t)s * 4 * g.GS + gl * g.GS + 4 * site) = acc;
      if (tape_out) {
        const int row = sm.row_chain[s];
        if (row >= 0) *(f32x4*)(tape_out + ((long long)row * 4 + gl) * g.GS + 4 * site) = acc;
      }
    }
  }
}

// One 16-channel convolution over the G resident samples: in -> out (LDS).  `wfrag` is the layer's
// fragment image ([K*K][64] f32x4, forward or transposed), `lo` the padding in front.
template <int K, int KW>
__device__ __forceinline__ void conv_layer(const ConvSmem& sm, const float* in, float* out,
                                           const ConvGeom& g, const float* wfrag, const float* bias16,
                                           int dir, int G, int ep, int wave, int lane,
                                           const float* tape_in, float* tape_out) {
  const int pl = lane & 15, gl = lane >> 4;
  f32x4 w[K * KW];
#pragma unroll
  for (int t = 0; t < K * KW; ++t) w[t] = *(const f32x4*)(wfrag + ((size_t)t * 64 + lane) * 4);
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (bias16) bias = *(const f32x4*)(bias16 + 4 * gl);
  const int n_pos = G * g.N, n_tiles = (n_pos + 15) >> 4;
  // Two adjacent position tiles at a time: two independent accumulator chains that share every
  // weight fragment.
  auto tiles = [&](int t0, auto nt_c) {
    constexpr int NTL = decltype(nt_c)::value;
    bool valid[NTL]; int a1[NTL], a2[NTL], sl[NTL];
    const char* base[NTL];
    int roff[NTL][K], coff[NTL][KW];
#pragma unroll
    for (int h = 0; h < NTL; ++h) {
      const int q = (t0 + h) * 16 + pl;
      valid[h] = q < n_pos;
      const unsigned info = sm.pinfo[valid[h] ? q : n_pos - 1];
      a2[h] = info & 1023; a1[h] = (info >> 10) & 1023; sl[h] = info >> 20;
      base[h] = (const char*)(in + (size_t)sl[h] * 4 * g.GS + gl * g.GS);
      const int* rt = sm.rtab + (dir * g.D1 + a1[h]) * 8;
      const int* ct = sm.ctab + (dir * g.D2 + a2[h]) * 8;
#pragma unroll
      for (int d = 0; d < K; ++d) roff[h][d] = rt[d];
#pragma unroll
      for (int d = 0; d < KW; ++d) coff[h][d] = ct[d];
    }
    f32x4 acc[NTL];
#pragma unroll
    for (int h = 0; h < NTL; ++h) acc[h] = bias;
    // B operands two taps ahead of the MFMAs that consume them (3-stage register ring per tile;
    // the sched_barrier keeps the compiler from sinking the reads back next to their use, which
    // would expose one LDS round trip per tap)
    f32x4 bq[NTL][3];
#pragma unroll
    for (int h = 0; h < NTL; ++h) {
      bq[h][0] = *(const f32x4*)(base[h] + roff[h][0] + coff[h][0]);
      if (K * KW > 1) bq[h][1] = *(const f32x4*)(base[h] + roff[h][K * KW > 1 ? 1 / KW : 0] + coff[h][K * KW > 1 ? 1 % KW : 0]);
    }
#pragma unroll
    for (int tap = 0; tap < K * KW; ++tap) {
      if (tap + 2 < K * KW) {
#pragma unroll
        for (int h = 0; h < NTL; ++h)
          bq[h][(tap + 2) % 3] = *(const f32x4*)(base[h] + roff[h][(tap + 2) / KW] + coff[h][(tap + 2) % KW]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int h = 0; h < NTL; ++h)
          acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[tap][e], bq[h][tap % 3][e], acc[h], 0, 0, 0);
    }
#pragma unroll
    for (int h = 0; h < NTL; ++h) {
      const int site = a1[h] * g.D2 + a2[h];
      float* dst = out + (size_t)sl[h] * 4 * g.GS + gl * g.GS + 4 * site;
      const int row = sm.row_chain[sl[h]];
      const unsigned trow = (unsigned)((row >= 0 ? row : 0) * 4 + gl) * (unsigned)g.GS + 4u * site;   // < 2^31: checked on the host
      f32x4 v = acc[h];
      if (ep == EP_ACT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = vmc_act_rt(g.hact, v[r]);
      } else if (ep == EP_SELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = selu_f(v[r]);
      } else if (ep == EP_RESADD || ep == EP_BACK_ADD) {
        const f32x4 old = *(const f32x4*)dst;
        v += old;
      } else if (ep == EP_BACK_DACT || ep == EP_BACK_SELU) {
        const f32x4 a = *(const f32x4*)(tape_in + trow);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          v[r] *= ep == EP_BACK_SELU ? selu_deriv_from_t(a[r]) : dact_from_a_rt(g.hact, a[r]);
      }
      if (valid[h]) {
        *(f32x4*)dst = v;
        if (tape_out && row >= 0) *(f32x4*)(tape_out + trow) = v;
      }
    }
  };
  // a wave's tiles in pairs (2w, 2w+1), (2w + 2 NW, ...); an odd last tile runs alone.  With more
  // than 25 taps (6 x 6: 144 weight registers) the pair's second accumulator set would spill, so
  // those kernels take the two tiles one after the other.
  constexpr bool PAIR = K * KW <= 25;
  for (int t0 = 2 * wave; t0 < n_tiles; t0 += 2 * CONV_WAVES) {
    if (PAIR && t0 + 1 < n_tiles) {
      tiles(t0, std::integral_constant<int, 2>{});
    } else {
      tiles(t0, std::integral_constant<int, 1>{});
      if (!PAIR && t0 + 1 < n_tiles) tiles(t0 + 1, std::integral_constant<int, 1>{});
    }
  }
}

// Whole forward of the G resident samples: spins (sm.xs) -> sm.red[s] = sum over sites and
// channels of the last feature map (wavefunctions.py:569, 760).  Barriers inside.
template <int K, int KW>
__device__ __forceinline__ void conv_forward(const ConvSmem& sm, const ConvGeom& g,
                                             const ConvParams& p, int G, int wave, int lane,
                                             float* tape, long long tape_stride) {
  float* last;
  if (!g.resnet) {
    // [Conv2dPeriodic, nonlinearity] x (n-1), Conv2dPeriodic            (wavefunctions.py:572-575)
    conv_first<K, KW>(sm, sm.buf0, g, p, G, g.n_conv > 1 ? EP_ACT : EP_LINEAR, wave, lane,
                  (tape && g.n_conv > 1) ? tape : nullptr, 0);
    __syncthreads();
    float* in = sm.buf0; float* out = sm.buf1;
    for (int l = 1; l < g.n_conv; ++l) {
      const bool is_last = l + 1 == g.n_conv;
      conv_layer<K, KW>(sm, in, out, g, p.wf + (size_t)(l - 1) * K * KW * 256, p.bias + 16 * l, 0, G,
                    is_last ? EP_LINEAR : EP_ACT, wave, lane, nullptr,
                    (tape && !is_last) ? tape + (long long)l * tape_stride : nullptr);
      __syncthreads();
      float* tmp = in; in = out; out = tmp;
    }
    last = in;
  } else {
    // initial_conv, then blocks h <- h + conv2(selu(conv1(h)))          (wavefunctions.py:766-772)
    // tape slot l-1 holds the input of convolution l: h before block k at slot 2k, selu(..) at 2k+1
    conv_first<K, KW>(sm, sm.buf0, g, p, G, EP_LINEAR, wave, lane, g.n_conv > 1 ? tape : nullptr, 0);
    __syncthreads();
    for (int l = 1; l + 1 < g.n_conv; l += 2) {
      conv_layer<K, KW>(sm, sm.buf0, sm.buf1, g, p.wf + (size_t)(l - 1) * K * KW * 256, p.bias + 16 * l,
                    0, G, EP_SELU, wave, lane, nullptr,
                    tape ? tape + (long long)l * tape_stride : nullptr);
      __syncthreads();
      conv_layer<K, KW>(sm, sm.buf1, sm.buf0, g, p.wf + (size_t)l * K * KW * 256, p.bias + 16 * (l + 1),
                    0, G, EP_RESADD, wave, lane, nullptr,
                    (tape && l + 2 < g.n_conv) ? tape + (long long)(l + 1) * tape_stride : nullptr);
      __syncthreads();
    }
    last = sm.buf0;
  }
  // fixed-order reduction: wave w sums samples w, w + 8, ...; padded channels hold exact zeros
  for (int s = wave; s < G; s += CONV_WAVES) {
    const float* m = last + (size_t)s * 4 * g.GS;
    float part = 0.f;
    for (int gq = 0; gq < 4; ++gq)
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
template <int K, int KW>
__global__ __launch_bounds__(CONV_THREADS, 2) void k_conv_rows(ConvRowsArgs a) {
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
    conv_forward<K, KW>(sm, g, a.p, G, wave, lane, tape, a.tape_stride);
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
template <int K, int KW>
__global__ __launch_bounds__(CONV_THREADS, 2) void k_conv_sweep(ConvSweepArgs a) {
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
    conv_forward<K, KW>(sm, g, a.p, G, wave, lane, nullptr, 0);
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
    conv_forward<K, KW>(sm, g, a.p, G, wave, lane, nullptr, 0);
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
template <int K, int KW>
__global__ __launch_bounds__(CONV_THREADS, 2) void k_conv_back(ConvBackArgs a) {
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
      float* d = sm.buf0 + (size_t)s * 4 * g.GS;
      for (int gq = 0; gq < 4; ++gq)
        for (int i = lane; i < 4 * g.N; i += 64) {
          const float v = (4 * gq + (i & 3)) < g.F ? sc : 0.f;
          d[gq * g.GS + i] = v;
          if (row >= 0) a.delta[(long long)(n - 1) * a.delta_stride + ((long long)row * 4 + gq) * g.GS + i] = v;
        }
    }
    __syncthreads();
    if (!g.resnet) {
      float* in = sm.buf0; float* out = sm.buf1;
      for (int l = n - 1; l >= 1; --l) {
        conv_layer<K, KW>(sm, in, out, g, a.p.wb + (size_t)(l - 1) * K * KW * 256, nullptr, 1, G,
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
            const float* d = sm.buf0 + (size_t)s * 4 * g.GS;
            if (row >= 0)
              for (int gq = 0; gq < 4; ++gq)
                for (int i = lane; i < 4 * g.N; i += 64)
                  a.delta[(long long)l * a.delta_stride + ((long long)row * 4 + gq) * g.GS + i] = d[gq * g.GS + i];
          }
        }
        conv_layer<K, KW>(sm, sm.buf0, sm.buf1, g, a.p.wb + (size_t)(l - 1) * K * KW * 256, nullptr, 1, G,
                      EP_BACK_SELU, wave, lane, a.tape + (long long)(l - 1) * a.tape_stride,
                      a.delta + (long long)(l - 1) * a.delta_stride);
        __syncthreads();
        conv_layer<K, KW>(sm, sm.buf1, sm.buf0, g, a.p.wb + (size_t)(l - 2) * K * KW * 256, nullptr, 1, G,
                      EP_BACK_ADD, wave, lane, nullptr, l == 2 ? a.delta : nullptr);
        __syncthreads();
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------- weight gradient
// sum_b (1 | w_b) * d logit_b / d W_l for every convolution: dW[tap][cin][cout] =
// sum_{b, pos} in_l[b, pos + tap, cin] * delta_l[b, pos, cout]  (and the bias: sum of delta_l).
// Grid (slice, layer): a workgroup walks the samples of its slice, each staged in LDS as
// [site][16 channels]; wave w owns the taps w, w + 8, ...; the reduction over positions is the k
// index of the MFMA (4 positions per instruction): A = input at the tap-shifted position (lane =
// cin), B = delta (lane = cout), and a second accumulator takes w_b * delta.  Partial sums go to
// ws[slice][layer]; k_conv_dw_reduce adds the slices in a fixed order into the accumulators.
#define DW_WAVES 8   // the weight-gradient kernel splits the taps over 8 waves (1 workgroup per CU)
template <int K, int KW>
__global__ __launch_bounds__(DW_WAVES * 64) void k_conv_dw(ConvDwArgs a) {
  constexpr int KK = K * KW;
  constexpr int TPW = (KK + DW_WAVES - 1) / DW_WAVES;        // taps per wave
  constexpr int T0 = (KK + 15) / 16;                             // tap tiles of the first layer
  extern __shared__ float s_dw[];
  const ConvGeom& g = a.g;
  const int l = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ml = lane & 15, gl = lane >> 4;
  const int Np = (g.N + 3) & ~3;
  float* s_in = s_dw;                       // [Np][16] (layer 0: [Np] spins)
  float* s_dl = s_dw + (size_t)Np * 16;     // [Np][16]
  const int per = (a.B + a.n_slices - 1) / a.n_slices;
  const int b0 = blockIdx.x * per, b1 = min(b0 + per, a.B);
  f32x4 acc1[TPW], acc2[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) { acc1[i] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[i] = acc1[i]; }
  f32x4 bacc1 = {0.f, 0.f, 0.f, 0.f}, bacc2 = bacc1;   // bias: A = ones (wave DW_WAVES-1)
  for (int i = threadIdx.x; i < (Np - g.N) * 16; i += blockDim.x) {   // zero the padded positions once
    s_dl[g.N * 16 + i] = 0.f;
    if (l > 0) s_in[g.N * 16 + i] = 0.f;
  }
  for (int b = b0; b < b1; ++b) {
    __syncthreads();
    const float* dsrc = a.delta + (long long)l * a.delta_stride + (long long)b * 4 * g.GS;
    for (int i = threadIdx.x; i < 4 * g.N; i += blockDim.x) {       // i = (group, site)
      const int gq = i / g.N, site = i - gq * g.N;
      *(f32x4*)(s_dl + site * 16 + 4 * gq) = *(const f32x4*)(dsrc + gq * g.GS + 4 * site);
    }
    if (l == 0) {
      for (int i = threadIdx.x; i < g.N; i += blockDim.x) s_in[i] = a.configs[(long long)b * g.N + i];
    } else {
      const float* isrc = a.tape + (long long)(l - 1) * a.tape_stride + (long long)b * 4 * g.GS;
      for (int i = threadIdx.x; i < 4 * g.N; i += blockDim.x) {
        const int gq = i / g.N, site = i - gq * g.N;
        *(f32x4*)(s_in + site * 16 + 4 * gq) = *(const f32x4*)(isrc + gq * g.GS + 4 * site);
      }
    }
    __syncthreads();
    const float wb = a.w[b];
    for (int c = 0; c < Np; c += 4) {
      const int pos = c + gl;
      const bool pv = pos < g.N;
      const float dv = s_dl[pos * 16 + ml];       // padded positions hold zeros
      const float dv2 = dv * wb;
      const int a1 = pos / g.D2, a2 = pos - a1 * g.D2;
      if (l == 0) {
        // A = spin at the tap-shifted position, lane m = tap 16 tt + m: tap tiles over waves
#pragma unroll
        for (int tt = 0; tt < T0; ++tt) {
          if ((tt % DW_WAVES) == wave) {
            int tap = 16 * tt + ml;
            tap = tap < KK ? tap : 0;
            const int n1 = wrap(a1 + tap / KW - g.lo, g.D1), n2 = wrap(a2 + tap % KW - g.lo2, g.D2);
            const float av = pv ? s_in[n1 * g.D2 + n2] : 0.f;
            acc1[tt / DW_WAVES] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, dv, acc1[tt / DW_WAVES], 0, 0, 0);
            acc2[tt / DW_WAVES] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, dv2, acc2[tt / DW_WAVES], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          const int tap = wave + i * DW_WAVES;       // wave-uniform
          if (tap < KK) {
            const int n1 = wrap(a1 + tap / KW - g.lo, g.D1), n2 = wrap(a2 + tap % KW - g.lo2, g.D2);
            const float av = pv ? s_in[(n1 * g.D2 + n2) * 16 + ml] : 0.f;
            acc1[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, dv, acc1[i], 0, 0, 0);
            acc2[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, dv2, acc2[i], 0, 0, 0);
          }
        }
      }
      if (wave == DW_WAVES - 1) {
        bacc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, dv, bacc1, 0, 0, 0);
        bacc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, dv2, bacc2, 0, 0, 0);
      }
    }
  }
  // partial sums: ws[slice][layer][2][(KK*16 + 1) * 16]: row (tap * 16 + cin) or KK*16 = bias, col cout
  const size_t rows = (size_t)KK * 16 + 1;
  float* w1 = a.ws + (((size_t)blockIdx.x * g.n_conv + l) * 2) * rows * 16;
  float* w2 = w1 + rows * 16;
  if (l == 0) {
#pragma unroll
    for (int tt = 0; tt < T0; ++tt)
      if ((tt % DW_WAVES) == wave)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int tap = 16 * tt + 4 * gl + r;        // accumulator row = tap
          if (tap < KK) {
            w1[(size_t)tap * 16 * 16 + ml] = acc1[tt / DW_WAVES][r];   // cin 0
            w2[(size_t)tap * 16 * 16 + ml] = acc2[tt / DW_WAVES][r];
          }
        }
  } else {
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int tap = wave + i * DW_WAVES;
      if (tap < KK)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          w1[((size_t)tap * 16 + 4 * gl + r) * 16 + ml] = acc1[i][r];    // row cin = 4g + r
          w2[((size_t)tap * 16 + 4 * gl + r) * 16 + ml] = acc2[i][r];
        }
    }
  }
  if (wave == DW_WAVES - 1 && gl == 0) {
    w1[(size_t)KK * 16 * 16 + ml] = bacc1[0];
    w2[(size_t)KK * 16 * 16 + ml] = bacc2[0];
  }
}

// g1 / g2 += sum over slices (fixed order) of the partial sums, scattered to the theta layout:
// convolution l: w [K][K][cin][F] then b [F]   (snt.Conv2D variable order)
__global__ void k_conv_dw_reduce(ConvDwArgs a, int KK) {
  const ConvGeom& g = a.g;
  const size_t rows = (size_t)KK * 16 + 1;
  long long off = 0;
  for (int l = 0; l < g.n_conv; ++l) {
    const int cin = l == 0 ? 1 : g.F;
    const long long nw = (long long)KK * cin * g.F, np = nw + g.F;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < np; i += (long long)gridDim.x * blockDim.x) {
      size_t src;
      if (i < nw) {
        const int co = (int)(i % g.F);
        const int ci = (int)((i / g.F) % cin);
        const int tap = (int)(i / ((long long)g.F * cin));
        src = ((size_t)tap * 16 + ci) * 16 + co;
      } else {
        src = (size_t)KK * 16 * 16 + (size_t)(i - nw);
      }
      float s1 = 0.f, s2 = 0.f;
      for (int sl = 0; sl < a.n_slices; ++sl) {
        const float* w1 = a.ws + (((size_t)sl * g.n_conv + l) * 2) * rows * 16;
        s1 += w1[src];
        s2 += w1[rows * 16 + src];
      }
      a.g1[off + i] += s1;
      a.g2[off + i] += s2;
    }
    off += np;
  }
}

// theta (snt.Conv2D order: per convolution w[K][K][cin][F], b[F]) -> fragment images
__global__ void k_conv_pack(const float* __restrict__ theta, ConvGeom g, float* w0, float* wf,
                            float* wb, float* bias) {
  const int KK = g.K * g.KW, Q0 = (KK + 3) / 4;
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long p0 = (long long)KK * g.F + g.F;                 // parameters of convolution 0
  const long long pl = (long long)KK * g.F * g.F + g.F;           // of every later one
  for (long long i = tid; i < (long long)Q0 * 64; i += stride) {
    const int q = (int)(i / 64), lane = (int)(i % 64), m = lane & 15, gq = lane >> 4;
    const int tap = 4 * q + gq;
    w0[i] = (tap < KK && m < g.F) ? theta[(long long)tap * g.F + m] : 0.f;
  }
  for (long long i = tid; i < (long long)g.n_conv * 16; i += stride) {
    const int l = (int)(i / 16), c = (int)(i % 16);
    const long long base = l == 0 ? (long long)KK * g.F : p0 + (long long)(l - 1) * pl + (long long)KK * g.F * g.F;
    bias[i] = c < g.F ? theta[base + c] : 0.f;
  }
  const long long per_layer = (long long)KK * 256;
  for (long long i = tid; i < (long long)(g.n_conv - 1) * per_layer; i += stride) {
    const int l = (int)(i / per_layer) + 1;
    const long long r = i % per_layer;
    const int tap = (int)(r / 256), lane = (int)((r % 256) / 4), e = (int)(r % 4);
    const int m = lane & 15, gq = lane >> 4;
    const float* w = theta + p0 + (long long)(l - 1) * pl;         // [tap][cin][cout]
    const int ci = 4 * gq + e, co = m;
    wf[i] = (ci < g.F && co < g.F) ? w[((long long)tap * g.F + ci) * g.F + co] : 0.f;
    // transposed convolution: output channel (lane m) = cin, k index = cout, taps flipped
    const int ci2 = m, co2 = 4 * gq + e, tap2 = KK - 1 - tap;
    wb[i] = (ci2 < g.F && co2 < g.F) ? w[((long long)tap2 * g.F + ci2) * g.F + co2] : 0.f;
  }
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
    case -2: { constexpr int KK_ = 2, KW_ = 1; CALL; } break;                           \
    case -3: { constexpr int KK_ = 3, KW_ = 1; CALL; } break;                           \
    case -4: { constexpr int KK_ = 4, KW_ = 1; CALL; } break;                           \
    case -5: { constexpr int KK_ = 5, KW_ = 1; CALL; } break;                           \
    case -6: { constexpr int KK_ = 6, KW_ = 1; CALL; } break;                           \
    default: return hipErrorInvalidValue;                                               \
  }

}  // namespace

// LDS budget of one workgroup: half a CU when a sample's feature maps allow two workgroups per CU
size_t conv_lds_cap(const ConvGeom& g) {
  static const int one_per_cu = getenv("CGS_VMC_CONV_WG_PER_CU") ? atoi(getenv("CGS_VMC_CONV_WG_PER_CU")) == 1 : 0;
  if (one_per_cu) return (size_t)160 * 1024;      // experiment: one workgroup with twice the samples
  return conv_rows_lds(g, 1) <= CONV_LDS_PER_WG ? (size_t)CONV_LDS_PER_WG : (size_t)160 * 1024;
}
int conv_waves() { return CONV_WAVES; }

size_t conv_rows_lds(const ConvGeom& g, int G) {
  const size_t xs = (size_t)((g.N + 3) & ~3);
  // buf0, buf1, xs, pinfo, row_chain, red + the sampler's cur_logit, prop, prop_u
  return ((size_t)G * 8 * g.GS + (size_t)G * xs + (size_t)G * g.N + (size_t)G * 7 + 16 * (size_t)(g.D1 + g.D2) + 16) * sizeof(float);
}

// samples per pass: the group size (<= 64, LDS <= 160 KiB) whose position tiles divide most evenly
// over the waves; ties go to the larger group
int conv_pick_group(const ConvGeom& g, int waves) {
  int best = 1; double best_eff = -1.0;
  for (int G = 1; G <= 64; ++G) {
    if (conv_rows_lds(g, G) > conv_lds_cap(g)) break;
    const int tiles = (G * g.N + 15) / 16;
    const int rounds = (tiles + waves - 1) / waves;
    const double eff = (double)G * g.N / 16.0 / ((double)rounds * waves);
    if (eff >= best_eff - 1e-9) { best = G; best_eff = eff; }
  }
  return best;
}

long long conv_num_params(int n_conv, int F, int taps) {
  const long long KK = taps;
  return KK * F + F + (long long)(n_conv - 1) * (KK * F * F + F);
}

hipError_t launch_conv_pack(hipStream_t s, const float* theta, const ConvGeom& g, float* w0,
                            float* wf, float* wb, float* bias) {
  hipLaunchKernelGGL(k_conv_pack, dim3(64), dim3(256), 0, s, theta, g, w0, wf, wb, bias);
  return hipGetLastError();
}

hipError_t launch_conv_rows(hipStream_t s, const ConvRowsArgs& a, int num_cus) {
  if (a.n_rows <= 0) return hipSuccess;
  const int groups = (a.n_rows + a.G - 1) / a.G;
  const size_t lds = conv_rows_lds(a.g, a.G);
  const int slots = num_cus * (lds <= CONV_LDS_PER_WG ? 2 : 1);      // co-resident workgroups
  const dim3 grid(groups < slots ? groups : slots);
  CONV_DISPATCH_K(a.g, return launch_k(k_conv_rows<KK_, KW_>, grid, lds, s, a));
  return hipSuccess;
}

hipError_t launch_conv_sweep(hipStream_t s, const ConvSweepArgs& a) {
  const dim3 grid((a.B + a.G - 1) / a.G);
  const size_t lds = conv_rows_lds(a.g, a.G);
  CONV_DISPATCH_K(a.g, return launch_k(k_conv_sweep<KK_, KW_>, grid, lds, s, a));
  return hipSuccess;
}

hipError_t launch_conv_back(hipStream_t s, const ConvBackArgs& a, int num_cus) {
  const int groups = (a.B + a.G - 1) / a.G;
  const size_t lds = conv_rows_lds(a.g, a.G);
  const int slots = num_cus * (lds <= CONV_LDS_PER_WG ? 2 : 1);
  const dim3 grid(groups < slots ? groups : slots);
  CONV_DISPATCH_K(a.g, return launch_k(k_conv_back<KK_, KW_>, grid, lds, s, a));
  return hipSuccess;
}

hipError_t launch_conv_dw(hipStream_t s, const ConvDwArgs& a) {
  const dim3 grid(a.n_slices, a.g.n_conv);
  const size_t lds = (size_t)((a.g.N + 3) & ~3) * 32 * sizeof(float);
  CONV_DISPATCH_K(a.g, {
    hipError_t e = launch_k(k_conv_dw<KK_, KW_>, grid, lds, s, a, DW_WAVES * 64);
    if (e != hipSuccess) return e;
  });
  hipLaunchKernelGGL(k_conv_dw_reduce, dim3(32), dim3(256), 0, s, a, a.g.K * a.g.KW);
  return hipGetLastError();
}
