// Persistent Metropolis exchange sampler k_sweep16 (graph_builders.py:38-89) and its launcher
// template.  Included by act_sweep.hip (one object per hidden activation and output epilogue: the
// instantiations dominate the build time) and by sweep_split.hip (the 3 x bf16 split experiment).
#pragma once
#include "common.hpp"
#include <cstdlib>
#include <type_traits>

#ifndef SWEEP_RT
#define SWEEP_RT 14   // resident k-tiles of the first H x H layer (sweep kernel)
#endif
#ifndef SWEEP_OCC0
#define SWEEP_OCC0 4    // waves per SIMD the all-streamed (RTP = 0) H = 256 variant is compiled for
#endif
#ifndef SWEEP_RT_WIDE
#define SWEEP_RT_WIDE 2      // resident k-tiles of the 384- and 512-unit variants
#endif
#ifndef SWEEP_SCALAR_NT
#define SWEEP_SCALAR_NT 16   // layers wider than this many tiles keep the weight base address scalar (see issue())
#endif
// The two switches below are on for the two-draw variants (UPRE = 2, N <= 128 sites) only: with
// five draws per lane (UPRE = 4) the second copy of layer 0 costs registers (6 -> 15 spilled);
// the split is also left out of the variants without W1 in LDS or without resident k-tiles (a few
// spills each).
#ifndef SWEEP_SPLIT
#define SWEEP_SPLIT(UPRE) ((UPRE) == 2)   // waves 4-7 run layer 0 without the Philox pieces
#endif
#ifndef SWEEP_PIN
#define SWEEP_PIN(UPRE) ((UPRE) == 2)     // keep every Philox round inside layer 0 (see pin_draw)
#endif
#ifndef SWEEP_HANDOFF
#define SWEEP_HANDOFF 1   // waves 4-7 draw the next step's uniforms in the serial phase (see HANDOFF)
#endif
#ifndef SWEEP_PF
#define SWEEP_PF 2    // stages of the weight prefetch ring; (16 - SWEEP_RT) % SWEEP_PF == 0
#endif


// ---- 3 x bf16 split pieces shared with tail_split.hip (EXPERIMENT, CGS_VMC_SPLIT_BF16): see there for the scheme
#ifndef VMC_SPLIT_HELPERS
#define VMC_SPLIT_HELPERS
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 sw_bf16x8;
typedef unsigned sw_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned sw_cvt_pk_bf16(float a, float b) {   // lo 16 bits = bf16(a), hi = bf16(b); round to nearest even
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void sw_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = sw_cvt_pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = sw_cvt_pk_bf16(r0, r1);
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
  l = sw_cvt_pk_bf16(s0, s1);
}
// the eight k slots of a lane for a 32-deep k-step: its registers of unit tiles 2 kt and 2 kt + 1
__device__ __forceinline__ void sw_split8(const f32x4& a, const f32x4& b, sw_u32x4& H, sw_u32x4& M, sw_u32x4& L) {
  unsigned h[4], m[4], l[4];
  sw_split2(a[0], a[1], h[0], m[0], l[0]);
  sw_split2(a[2], a[3], h[1], m[1], l[1]);
  sw_split2(b[0], b[1], h[2], m[2], l[2]);
  sw_split2(b[2], b[3], h[3], m[3], l[3]);
  H = sw_u32x4{h[0], h[1], h[2], h[3]};
  M = sw_u32x4{m[0], m[1], m[2], m[3]};
  L = sw_u32x4{l[0], l[1], l[2], l[3]};
}
__device__ __forceinline__ f32x4 sw_mfma(const sw_u32x4& a, const sw_u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sw_bf16x8, a), __builtin_bit_cast(sw_bf16x8, b), c, 0, 0, 0);
}
struct SwFrag { sw_u32x4 h, m, l; };
#endif

// --------------------------------------------------------------------------------- sweep16
// One workgroup (NW = 4 or 8 waves) owns 16 chains for the whole launch.  Per mc_step:
//   proposals (Philox, argmax/argmin of s*u over sites: graph_builders.py:59-65)
//   z1' = z1 + 2 (W1[i_dn] - W1[i_up])            (rank-2 form of the forward at :74)
//   layers 2..L by 16x16x4 MFMA, wave w owns output units [w*Hp/NW, (w+1)*Hp/NW)
//   accept = exp(logit' - logit) > sqrt(u)        (graph_builders.py:75-79), evaluated as
//            logit' - logit > 0.5 log u by the wave that owns the chain
// Chain state (spins, logit) stays in LDS, the committed z1 of a thread's columns in its registers
// (W1-in-LDS variants); z1 and logit are recomputed from the spins at launch start and end so the
// cache written back never carries incremental drift.
// With 8 waves the chains belong to waves 0-3; waves 4-7 use the time until barrier0 to draw the
// next step's Philox uniforms and hand them over as sortable keys (HANDOFF below), so that the
// MFMA phases carry no VALU work.
//
// Weight traffic: the A-operand fragments of the first RT (= 14 of 16 at H = 256) k-tiles of
// the FIRST H x H layer of this wave's output units are loaded once and stay in registers for
// the whole launch; everything else streams from L2 through a PF-stage register ring whose
// loads are issued PF-1 k-tiles ahead of their use and run across layer boundaries.
// 16-lane (DPP row) all-reduce steps: lane^1, lane^2 (quad_perm), then row_half_mirror and
// row_mirror, which combine the already-uniform quads / halves.  Pure VALU, no LDS crossbar.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false);
}
#define DPP_XOR1 0xB1
#define DPP_XOR2 0x4E
#define DPP_HALF_MIRROR 0x141
#define DPP_MIRROR 0x140

// s_memtime stamp for the diagnostic instantiation (STAMP = true) only; the production kernel
// (STAMP = false) executes none of it.
__device__ __forceinline__ unsigned long long vmc_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

// W1L: the first-layer matrix W1 [N][Hp] also lives in LDS (when it fits), which turns the
// rank-2 gather of two W1 rows per chain and step from an L2 round trip into LDS reads; the
// accepted move is then folded into z1 lazily at the start of the next step instead of being
// double-buffered.
// FAST: the production path (Philox draws prefetched, no injected proposals, no debug dump);
// the general instantiation keeps every path.
// NW: waves per workgroup (4 or 8).  With 8 waves (2 per SIMD) each wave owns NT/8 output
// tiles; the two co-resident waves hide each other's vmcnt / LDS stalls inside the layers.
// Proposals, accept and the Philox chains stay on waves 0-3 (4 chains each).
// RBM: RestrictedBoltzmannNetwork epilogue: the last layer's units go through log cosh (its
// output "dot" is against ones) and the onsite term x . w_on is tracked per chain in LDS.
// RTP: k-tiles of the first H x H layer whose fragments stay in registers (14 with 8 waves; the
// 4-wave variant of H = 256 owns 4 output tiles per wave and keeps fewer so that it fits 256
// registers, i.e. two workgroups per CU).
// UPRE: Philox site blocks per lane drawn one step ahead (2: N <= 128 sites, 4: N <= 256).
// ACT: hidden activation (layers.NONLINEARITIES id); relu is the tuned path.
// SW (EXPERIMENT, round 5, CGS_VMC_SPLIT_BF16=2): the H x H layers on the BF16 matrix cores with every f32 operand
// as three bf16 terms and six products accumulated in fp32 (tail_split.hip has the scheme and the slot order of the
// weight image p16s).  256 units, 8 waves, fully_connected, relu, W1 in L2 only.  Wave w owns unit tiles 2 w and
// 2 w + 1 -- exactly the two tiles of the 32-deep k-step w of the NEXT layer -- so it splits its own activations
// once in its epilogue and leaves them in LDS as that k-step's B operand (hi, mid, lo); the operand buffers are
// [8 k-steps][3 terms][64 lanes][16 bytes] = 24 KiB each.  RTP / 2 k-steps of layer 0 stay in registers, the rest
// streams L2 -> registers through a two-item ring (an item = the six 1 KiB fragments of one k-step and both tiles).
template <int NT, int NW, int RTP, bool STAMP, bool W1L, bool FAST, int UPRE, bool RBM, int ACT, bool SW = false>
__device__ __forceinline__ void sweep16_body(const SweepArgs& a) {
  static_assert(NT % NW == 0, "output tiles must divide over the waves");
  static_assert(!SW || (NT == 16 && NW == 8 && !RBM && !W1L && ACT == VMC_ACT_RELU_ && !STAMP), "split sampler: 256 relu units, 8 waves, W1 in L2");
  constexpr int NTH = NW * 64;
  constexpr int Hp = NT * 16, TO = NT / NW, ZS = Hp + 4, W1S = Hp + 4, PF = SWEEP_PF;
  constexpr int XB = SW ? 8 * 3 * 256 : NT * 256;   // floats per operand buffer
  constexpr int RT = NT < RTP ? NT : RTP;   // k-tiles of the first H x H layer kept in registers
  static_assert((NT - RT) % PF == 0, "the streamed k-tiles of layer 0 must fill whole ring turns");
  extern __shared__ float smem[];
  const int N = a.N, Nst = (N + 3) & ~3;
  float* s_spin = smem;                       // [16][Nst]
  float* s_z1 = s_spin + 16 * Nst;            // [W1L ? 1 : 2][16][ZS]
  float* s_x = s_z1 + (W1L ? 1 : 2) * 16 * ZS;  // [2][NT][64][4]
  float* s_logit = s_x + 2 * XB;              // [16]
  float* s_u = s_logit + 16;                  // [16]
  int* s_iup = (int*)(s_u + 16);              // [16]
  int* s_idn = s_iup + 16;                    // [16]
  int* s_sel = s_idn + 16;                    // [16]
  int* s_pacc = s_sel + 48;                   // [16] previous step's accept flag (after two unused [16] slots)
  float* s_hlu = (float*)(s_pacc + 16);       // [16] 0.5 log(u_accept)
  float* s_wout = s_hlu + 16;                 // [Hp]
  float* s_bias = s_wout + Hp;                // [n_hidden][Hp] biases of the H x H layers
  // RBM onsite weights: with W1 in LDS they live in the first padding float of each W1 row
  // (s_w1[n * W1S + Hp]), otherwise in their own [Nst] array
  float* s_won = s_bias + a.n_hidden * Hp;
  float* s_on = s_won + ((RBM && !W1L) ? Nst : 0);   // [16] x . w_on of the committed chains (RBM only)
  float* s_w1 = s_on + (RBM ? 16 : 0);        // [N][W1S] (W1L only)
  auto won_at = [&](int n) { return W1L ? s_w1[n * W1S + Hp] : s_won[n]; };

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform -> SGPR
  const int g = lane >> 4, j = lane & 15;
  const int chain0 = blockIdx.x * 16;
  const PackedParams& pp = a.pp;
  const int n_hidden = a.n_hidden;
  // [4 waves][16 chains] partials of the output dot live in the operand buffer the LAST layer does
  // not read (free from the barrier at that layer's top until the next step's build, which
  // comes after every reader of s_part has passed barrier0)
  float* s_part = s_x + (n_hidden == 0 ? 1 : (n_hidden & 1)) * XB;

  for (int i = tid; i < 16 * Nst; i += NTH) {
    const int c = i / Nst, n = i % Nst, gc = chain0 + c;
    float v = 0.f;
    if (n < N) v = gc < a.B ? a.configs_in[(long long)gc * N + n] : ((n & 1) ? -1.f : 1.f);
    s_spin[i] = v;
  }
  if (tid < 16) { s_sel[tid] = 0; s_pacc[tid] = 0; }
  for (int i = tid; i < n_hidden * Hp; i += NTH) s_bias[i] = pp.bh[i];
  for (int i = tid; i < Hp; i += NTH) s_wout[i] = pp.woutp[i];
  if (RBM) {
    if (W1L) { for (int i = tid; i < N; i += NTH) s_w1[i * W1S + Hp] = pp.won[i]; }
    else { for (int i = tid; i < Nst; i += NTH) s_won[i] = i < N ? pp.won[i] : 0.f; }
  }
  if (W1L) {   // 8 loads in flight per thread: the copy costs one L2 round trip per 8 vectors
    const int total = N * (Hp / 4);
    for (int base = tid; base < total; base += 8 * NTH) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = min(base + u * NTH, total - 1);
        v[u] = *(const f32x4*)(pp.w1p + (long long)(i / (Hp / 4)) * Hp + 4 * (i % (Hp / 4)));
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * NTH;
        if (i < total) *(f32x4*)(s_w1 + (i / (Hp / 4)) * W1S + 4 * (i % (Hp / 4))) = v[u];
      }
    }
  }
  const float bout = pp.bout[0];

  // register-resident fragments of the first H x H layer
  f32x4 wres[RT * TO > 0 ? RT * TO : 1];
  constexpr int RES = SW ? RT / 2 : 0;           // resident 32-deep k-steps of layer 0 (split sampler)
  SwFrag wres_s[RES > 0 ? RES : 1][TO];
  typedef const __attribute__((address_space(1))) sw_u32x4* sw_gp;
  // (the image base is re-made opaque once per layer pass: with a visible base the compiler hoists a 64-bit scalar
  // address per (layer, k-step, tile) out of the step loop and spills them)
  const unsigned* sw_base = a.p16s + (long long)wave * TO * 8 * 3 * 256;
  auto sw_frag = [&](int l, int kt, int to) {    // p16s: [layer][unit tile 16][k-step 8][term 3][64 lanes][4 dwords]
    sw_gp q = (sw_gp)(sw_base + (((long long)l * NT + to) * 8 + kt) * 3 * 256) + lane;
    SwFrag f;
    f.h = q[0]; f.m = q[64]; f.l = q[128];
    return f;
  };
  if (SW && n_hidden > 0) {
#pragma unroll
    for (int kt = 0; kt < RES; ++kt)
#pragma unroll
      for (int to = 0; to < TO; ++to) wres_s[kt][to] = sw_frag(0, kt, to);
  }
  if (!SW && n_hidden > 0) {
    const f32x4* __restrict__ wp0 = (const f32x4*)pp.p16 + wave * TO * NT * 64;
#pragma unroll
    for (int ti = 0; ti < RT; ++ti)
#pragma unroll
      for (int to = 0; to < TO; ++to)
        wres[ti * TO + to] = wp0[(to * NT + ti) * 64 + lane];
  }
  __syncthreads();

  // z1 from the spins (first layer, exact): thread -> (column, chain group)
  auto z1_direct = [&]() {
    constexpr int GROUPS = NTH / Hp > 0 ? NTH / Hp : 1;   // chain groups when Hp < NTH
    constexpr int CPG = 16 / GROUPS;
    const int col = tid % Hp, grp = tid / Hp;
    if (grp < GROUPS) {
      float acc[CPG];
      const float b = pp.b1p[col];
#pragma unroll
      for (int c = 0; c < CPG; ++c) acc[c] = b;
      if (W1L) {   // W1 is LDS-resident: no global round trip per site
        for (int n = 0; n < N; ++n) {
          const float w = s_w1[n * W1S + col];
#pragma unroll
          for (int c = 0; c < CPG; ++c) acc[c] = fmaf(s_spin[(grp * CPG + c) * Nst + n], w, acc[c]);
        }
      } else {     // stream W1 from L2 eight rows at a time (all eight loads in flight together)
        int n = 0;
        for (; n + 8 <= N; n += 8) {
          float w[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) w[u] = pp.w1p[(long long)(n + u) * Hp + col];
#pragma unroll
          for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CPG; ++c) acc[c] = fmaf(s_spin[(grp * CPG + c) * Nst + n + u], w[u], acc[c]);
        }
        for (; n < N; ++n) {
          const float w = pp.w1p[(long long)n * Hp + col];
#pragma unroll
          for (int c = 0; c < CPG; ++c) acc[c] = fmaf(s_spin[(grp * CPG + c) * Nst + n], w, acc[c]);
        }
      }
#pragma unroll
      for (int c = 0; c < CPG; ++c) s_z1[(grp * CPG + c) * ZS + col] = acc[c];
    }
  };
  // exact onsite term of the committed spins (RBM): one thread per chain
  auto onsite_direct = [&]() {
    if (RBM && tid < 16) {
      float acc = 0.f;
      for (int n = 0; n < N; ++n) acc = fmaf(s_spin[tid * Nst + n], won_at(n), acc);
      s_on[tid] = acc;
    }
  };

  const uint2 key = make_uint2(a.seed_lo, a.seed_hi);
  const int nblk = (N + 3) >> 2;

  // ---- proposals (graph_builders.py:59-65).  Lane (c = 4*wave + g, sub = j) owns the site
  // blocks sub, sub+16, ... of chain c.  The Philox draws of step t+1 do not depend on the
  // chain state, so they are computed one step ahead inside the MFMA phase of step t (UPRE
  // blocks per lane, i.e. N <= 64*UPRE/... sites); only the argmax/argmin of s*u is left for
  // the start of the step.
  // ND draws per lane and step: the UPRE site blocks j, j+16, ... and the acceptance block.  With
  // UPRE = 2 the acceptance draw rides in lane 15's second slot when that block is beyond the
  // lattice (N <= 124), so ND = 2; with UPRE = 4 it has a slot of its own (ND = 5).
  constexpr int ND = UPRE == 2 ? 2 : UPRE + 1;
  const bool use_pref = FAST || ((nblk <= 16 * UPRE) && (a.inj_up == nullptr));
  // chains 4w..4w+3 belong to wave w (< 4); waves 4-7 of an 8-wave workgroup see their SIMD
  // partner's chains (only the hand-over draws below use that)
  const int my_c = (NW > 4 ? (wave & 3) : wave) * 4 + g;
  const uint32_t my_gid = (uint32_t)(a.chain_offset + chain0 + my_c);
  float u_pre[4 * UPRE];
  float u_pre_acc = 0.f;
#pragma unroll
  for (int i = 0; i < 4 * UPRE; ++i) u_pre[i] = 0.f;

  const bool acc_in_b = (15 + 16 >= nblk);          // UPRE = 2: lane 15's second slot is free
  auto ctr_of = [&](int d, unsigned long long step) {
    uint32_t blk = (uint32_t)(j + 16 * d);
    if (d >= UPRE) blk = VMC_ACCEPT_BLOCK;
    else if (ND == UPRE && d == UPRE - 1 && j + 16 * d >= nblk) blk = VMC_ACCEPT_BLOCK;
    return make_uint4(blk, my_gid, (uint32_t)step, (uint32_t)(step >> 32));
  };
  auto finish_draw = [&](const uint4 (&cs)[ND], unsigned long long step) {
#pragma unroll
    for (int d = 0; d < UPRE; ++d) {
      u_pre[4 * d + 0] = u32_to_uniform(cs[d].x); u_pre[4 * d + 1] = u32_to_uniform(cs[d].y);
      u_pre[4 * d + 2] = u32_to_uniform(cs[d].z); u_pre[4 * d + 3] = u32_to_uniform(cs[d].w);
    }
    if (ND > UPRE) {
      u_pre_acc = u32_to_uniform(cs[ND - 1].x);
    } else if (acc_in_b) {
      u_pre_acc = u_pre[4 * (UPRE - 1)];
    } else {
      const uint4 r = philox4x32_10(
          make_uint4(VMC_ACCEPT_BLOCK, my_gid, (uint32_t)step, (uint32_t)(step >> 32)), key);
      u_pre_acc = u32_to_uniform(r.x);
    }
  };
  auto draw_all = [&](unsigned long long step) {     // un-overlapped form
    uint4 cs[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) cs[d] = philox4x32_10(ctr_of(d, step), key);
    finish_draw(cs, step);
  };

  // branch-free (v_cndmask) argmax / argmin combine with the first-index tie rule of
  // tf.argmax / tf.argmin
  auto reduce_and_publish = [&](float vmax, int imax, float vmin, int imin, float uacc) {
#define VMC_RED_STEP(CTRL)                                                            \
    {                                                                                 \
      const float ov = dpp_f<CTRL>(vmax); const int oi = dpp_i<CTRL>(imax);           \
      const bool tmax = (ov > vmax) | ((ov == vmax) & (oi < imax));                   \
      vmax = tmax ? ov : vmax; imax = tmax ? oi : imax;                               \
      const float pv = dpp_f<CTRL>(vmin); const int pi = dpp_i<CTRL>(imin);           \
      const bool tmin = (pv < vmin) | ((pv == vmin) & (pi < imin));                   \
      vmin = tmin ? pv : vmin; imin = tmin ? pi : imin;                               \
    }
    VMC_RED_STEP(DPP_XOR1) VMC_RED_STEP(DPP_XOR2) VMC_RED_STEP(DPP_HALF_MIRROR) VMC_RED_STEP(DPP_MIRROR)
#undef VMC_RED_STEP
    if (j == 15) {
      s_iup[my_c] = imax;   // argmax of s*u: the UP spin to lower   (graph_builders.py:64-65)
      s_idn[my_c] = imin;   // argmin of s*u: the DOWN spin to raise (graph_builders.py:62-63)
      s_u[my_c] = uacc;
      s_hlu[my_c] = 0.5f * __logf(uacc);
    }
  };
  // HANDOFF (every prefetching 8-wave variant with resident k-tiles): the NEXT step's uniforms are drawn
  // by waves 4-7 at the top of the iteration -- while waves 0-3 resolve the previous step and
  // build the proposals, they have nothing to do until barrier0 -- and handed over through the
  // part of operand buffer 1 that is dead between the output dot and layer 0's epilogue
  // (behind s_part when that lives there).  Layer 0 then carries no Philox pieces at all.
  // The area holds UPRE + 1 float4 slots per lane of waves 0-3; where it does not fit operand
  // buffer 1 (256 units with 129..256 sites: five slots) it is a region of its own behind the
  // chain state, which the launcher grants when the lattice leaves room (a.uh_lds; those shapes
  // do not have W1 in LDS).
  constexpr int UH_FLOATS = 4 * (UPRE + 1) * 256;
  constexpr bool UH_IN_X = NW * 16 + UH_FLOATS <= XB;
  constexpr bool HANDOFF_T = SWEEP_HANDOFF && FAST && NW == 8 && RT > 0 && (UH_IN_X || !W1L);
  const bool handoff = HANDOFF_T && n_hidden > 0 && (UH_IN_X || a.uh_lds != 0);
  float* s_uh = UH_IN_X ? s_x + XB + NW * 16 : s_w1;    // [4 waves][UPRE + 1][64 lanes][4]
  // validity of this lane's 4*UPRE prefetched sites (bit k <-> site 4*(j+16*(k/4)) + k%4)
  unsigned pre_valid = 0;
#pragma unroll
  for (int k = 0; k < 4 * UPRE; ++k)
    if (4 * (j + 16 * (k / 4)) + (k % 4) < N) pre_valid |= 1u << k;

  // Hand-over variants carry each site's uniform as a sortable key: (24-bit draw << 8) | (255 - site),
  // 0 for a site beyond the lattice.  The larger key is the larger uniform and, among equal
  // uniforms, the smaller site index -- the first-index tie rule of tf.argmax / tf.argmin
  // (graph_builders.py:62-65) -- so argmax / argmin of s*u over the up / down spins are two
  // integer max reductions.  (A draw of exactly 0 at site 255 is indistinguishable from "no site";
  // it would have to be the largest draw among all up or all down spins to matter.)
  auto to_keys = [&]() {
#pragma unroll
    for (int k = 0; k < 4 * UPRE; ++k) {
      const int n = 4 * (j + 16 * (k / 4)) + (k % 4);
      const unsigned kk = (unsigned)(u_pre[k] * 16777216.f);
      const unsigned key = ((pre_valid >> k) & 1u) ? ((kk << 8) | (unsigned)(255 - n)) : 0u;
      u_pre[k] = __uint_as_float(key);
    }
  };
  // proposals of absolute step `step` into s_iup / s_idn / s_u
  auto proposals = [&](unsigned long long step) {
    if (NW > 4 && wave >= 4) return;   // chains 4w..4w+3 belong to waves 0-3
    if (!FAST && a.inj_up) {
      if (tid < 16) {
        const int gc = chain0 + tid;
        const bool ok = gc < a.B;
        s_iup[tid] = ok ? a.inj_up[gc] : 0;
        s_idn[tid] = ok ? a.inj_dn[gc] : 1;
        s_u[tid] = ok ? a.inj_u[gc] : 2.f;
        s_hlu[tid] = 0.5f * __logf(s_u[tid]);
      }
      return;
    }
    if (HANDOFF_T && handoff) {   // keys handed over by waves 4-7 (or made from the launch's first draw)
      unsigned kup = 0u, kdn = 0u;
#pragma unroll
      for (int b = 0; b < UPRE; ++b) {
        const f32x4 sp = *(const f32x4*)(s_spin + my_c * Nst + 4 * (j + 16 * b));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned key = __float_as_uint(u_pre[4 * b + e]);
          kup = max(kup, sp[e] > 0.f ? key : 0u);
          kdn = max(kdn, sp[e] < 0.f ? key : 0u);
        }
      }
#define VMC_KEY_STEP(CTRL) \
      kup = max(kup, (unsigned)dpp_i<CTRL>((int)kup)); kdn = max(kdn, (unsigned)dpp_i<CTRL>((int)kdn));
      VMC_KEY_STEP(DPP_XOR1) VMC_KEY_STEP(DPP_XOR2) VMC_KEY_STEP(DPP_HALF_MIRROR) VMC_KEY_STEP(DPP_MIRROR)
#undef VMC_KEY_STEP
      if (j == 15) {
        // A chain without an up (or without a down) spin has no exchange move: no key survives the
        // reduction and 255 - (0 & 255) would address site 255 of a smaller lattice (W1 rows, a
        // neighbouring chain's spins).  Such a chain proposes the null move 0 <-> 0 with a NaN
        // acceptance uniform, which every accept test rejects (x > NaN is false): it stays frozen.
        // The reference's scatter arithmetic (graph_builders.py:67-71) would write a spin of +-3.
        const bool none = (kup == 0u) | (kdn == 0u);
        const float ua = none ? __uint_as_float(0x7fc00000u) : u_pre_acc;
        s_iup[my_c] = none ? 0 : 255 - (int)(kup & 255u);   // argmax of s*u: the UP spin to lower   (graph_builders.py:64-65)
        s_idn[my_c] = none ? 0 : 255 - (int)(kdn & 255u);   // argmin of s*u: the DOWN spin to raise (graph_builders.py:62-63)
        s_u[my_c] = ua;
        s_hlu[my_c] = 0.5f * __logf(ua);
      }
      return;
    }
    float vmax = -3.f, vmin = 3.f;
    int imax = 0x7fffffff, imin = 0x7fffffff;
    if (use_pref) {       // uniforms were drawn during the previous step's MFMA phase
      // branch-free: out-of-lattice sites read (valid LDS) garbage and are masked to -3 / +3
#pragma unroll
      for (int b = 0; b < UPRE; ++b) {
        const int blk = j + 16 * b;
        const f32x4 sp = *(const f32x4*)(s_spin + my_c * Nst + 4 * blk);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = 4 * blk + e;
          const bool ok = (pre_valid >> (4 * b + e)) & 1u;
          const float v = sp[e] * u_pre[4 * b + e];
          const float vx = ok ? v : -3.f, vn = ok ? v : 3.f;
          const bool tmax = vx > vmax, tmin = vn < vmin;
          vmax = tmax ? vx : vmax; imax = tmax ? n : imax;
          vmin = tmin ? vn : vmin; imin = tmin ? n : imin;
        }
      }
      reduce_and_publish(vmax, imax, vmin, imin, u_pre_acc);
      return;
    }
    if (FAST) return;
    for (int blk = j; blk < nblk; blk += 16) {
      const uint4 r = philox4x32_10(
          make_uint4((uint32_t)blk, my_gid, (uint32_t)step, (uint32_t)(step >> 32)), key);
      const f32x4 sp = *(const f32x4*)(s_spin + my_c * Nst + 4 * blk);
      const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = 4 * blk + e;
        if (n < N) {
          const float v = sp[e] * u32_to_uniform(rr[e]);
          if (v > vmax) { vmax = v; imax = n; }
          if (v < vmin) { vmin = v; imin = n; }
        }
      }
    }
    float uacc = 0.f;
    if (j == 15) {   // the lane with the fewest site blocks also draws the acceptance uniform
      const uint4 r = philox4x32_10(
          make_uint4(VMC_ACCEPT_BLOCK, my_gid, (uint32_t)step, (uint32_t)(step >> 32)), key);
      uacc = u32_to_uniform(r.x);
    }
    reduce_and_publish(vmax, imax, vmin, imin, uacc);
  };

  if (!FAST && a.dbg_up != nullptr) {   // debug_proposals: dump the draw of step0, do not move
    if (use_pref) draw_all(a.step0);
    proposals(a.step0);
    __syncthreads();
    if (tid < 16 && chain0 + tid < a.B) {
      a.dbg_up[chain0 + tid] = s_iup[tid];
      a.dbg_dn[chain0 + tid] = s_idn[tid];
      a.dbg_u[chain0 + tid] = s_u[tid];
    }
    return;
  }

  f32x4 own[TO];  // relu'd activations of this wave's own output tiles (B-operand layout)
  f32x4 zlast[TO];  // RBM: pre-activations of the last layer (the gradient path wants tanh of them)
  f32x4 zown[TO];   // cosine: pre-activations of `own` (the gradient path wants -sin of them)

  // diagnostic stamps (STAMP instantiation only)
  unsigned long long cyc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;
  bool stamp_on = false;
#define SWEEP_STAMP(k) \
  if (STAMP) { const unsigned long long t1_ = vmc_stamp(); if (stamp_on) cyc[k] += t1_ - t0; t0 = t1_; }

  // builds the layer-2 input operand (and the candidate z1 when with_delta)
  bool save_acts = false;   // final refresh: also write the activations for the gradient path
  auto save_own = [&](int l) {
    if (chain0 + j < a.B) {
      float* dst = a.act_out + ((long long)l * a.B + chain0 + j) * Hp;
#pragma unroll
      for (int to = 0; to < TO; ++to) {
        f32x4 v = own[to];
        if (RBM && l == n_hidden) {   // d sum log cosh(z) / d z = tanh(z): the last layer's delta
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = tanhf(zlast[to][e]);
        }
        *(f32x4*)(dst + 16 * (wave * TO + to) + 4 * g) = v;
        if (ACT == VMC_ACT_COS_ && a.dact_out && !(RBM && l == n_hidden)) {   // f'(z) = -sin z
          f32x4 d;
#pragma unroll
          for (int e = 0; e < 4; ++e) d[e] = -__sinf(zown[to][e]);
          *(f32x4*)(a.dact_out + ((long long)l * a.B + chain0 + j) * Hp + 16 * (wave * TO + to) + 4 * g) = d;
        }
      }
    }
  };
  // last-stage activation: relu (FC; the dot with w_out follows) or log cosh (RBM, w_out = 1)
  auto finish_own = [&](int to, const f32x4& z, bool last) {
    if (ACT == VMC_ACT_COS_) zown[to] = z;
#pragma unroll
    for (int e = 0; e < 4; ++e) own[to][e] = vmc_act<ACT>(z[e]);
    if (RBM && last) {
      zlast[to] = z;
#pragma unroll
      for (int e = 0; e < 4; ++e) own[to][e] = vmc_logcosh(z[e]);
    }
  };
  // split sampler: this wave's two unit tiles = k-step `wave` of the next layer's B operand, as three bf16 terms
  auto sw_publish = [&](float* buf) {
    if constexpr (SW) {
      sw_u32x4 H, M, L;
      sw_split8(own[0], own[TO - 1], H, M, L);
      sw_u32x4* dst = (sw_u32x4*)buf + (wave * 3) * 64 + lane;
      dst[0] = H; dst[64] = M; dst[128] = L;
    }
  };
  f32x4 zreg[TO];    // committed z1 of this thread's columns (W1L build)
  bool z_valid = false;
  f32x4 dprev[TO];   // W1[i_dn] - W1[i_up] of the previous proposal (W1L build)
#pragma unroll
  for (int to = 0; to < TO; ++to) dprev[to] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto build = [&](bool with_delta) {
    if (W1L) {
      // single z1 buffer; the previous step's accepted move is folded in first (each thread
      // owns fixed elements of z1, so no barrier is needed for the read-modify-write).
      // Branch-free and with every LDS read issued before the first use: the phase costs one
      // LDS round trip instead of three per output tile.
      // The difference of the two W1 rows of the previous proposal (dprev, in registers) is what an
      // accepted move adds: no second pair of row reads.
      float* zrow = s_z1 + j * ZS;
      const float cp = s_pacc[j] != 0 ? 2.f : 0.f;
      const float cd = with_delta ? 2.f : 0.f;
      const float* wa = s_w1 + (with_delta ? s_idn[j] : 0) * W1S;
      const float* wb = s_w1 + (with_delta ? s_iup[j] : 0) * W1S;
      // The committed z1 of this thread's columns lives in registers (zreg) between steps; LDS
      // holds it only where another phase produces or consumes it: the launch's cache load, the
      // exact refresh passes (z1_direct) and the final write-back, which follows a refresh.
      const bool from_lds = !with_delta || !z_valid;
      f32x4 x1[TO], y1[TO];
#pragma unroll
      for (int to = 0; to < TO; ++to) {
        const int col = 16 * (wave * TO + to) + 4 * g;
        if (from_lds) zreg[to] = *(const f32x4*)(zrow + col);
        x1[to] = *(const f32x4*)(wa + col); y1[to] = *(const f32x4*)(wb + col);
      }
      z_valid = true;
#pragma unroll
      for (int to = 0; to < TO; ++to) {
        const int t = wave * TO + to;
        f32x4 zc;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          zreg[to][e] = fmaf(cp, dprev[to][e], zreg[to][e]);        // committed z1
          dprev[to][e] = x1[to][e] - y1[to][e];
          zc[e] = fmaf(cd, dprev[to][e], zreg[to][e]);              // candidate z1'
        }
        finish_own(to, zc, n_hidden == 0);
        *(f32x4*)(s_x + (t * 64 + lane) * 4) = own[to];
      }
      if (save_acts) save_own(0);
      return;
    }
    const int sel = s_sel[j];
    const float* zc = s_z1 + sel * 16 * ZS + j * ZS;
    float* zn = s_z1 + (sel ^ 1) * 16 * ZS + j * ZS;
    const float* wa = pp.w1p;
    const float* wb = pp.w1p;
    if (with_delta) {
      wa = pp.w1p + (long long)s_idn[j] * Hp;
      wb = pp.w1p + (long long)s_iup[j] * Hp;
    }
#pragma unroll
    for (int to = 0; to < TO; ++to) {
      const int t = wave * TO + to, col = 16 * t + 4 * g;
      f32x4 z = *(const f32x4*)(zc + col);
      if (with_delta) {
        const f32x4 x = *(const f32x4*)(wa + col);
        const f32x4 y = *(const f32x4*)(wb + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) z[e] = fmaf(2.f, x[e] - y[e], z[e]);
        *(f32x4*)(zn + col) = z;
      }
      finish_own(to, z, n_hidden == 0);
      if (!SW) *(f32x4*)(s_x + (t * 64 + lane) * 4) = own[to];
    }
    if constexpr (SW) sw_publish(s_x);
    if (save_acts) save_own(0);
  };

  // layers 2..L + output dot; leaves per-wave partial logits in s_part.
  // Weight stream = [(layer 0, ti = RT..NT-1), (layer 1, all ti), ...] through a PF-stage
  // register ring; item q lives in stage q % PF and is issued PF-1 items ahead of its use.
  // The ring lives across mc_steps: the last PF-1 issues of a pass fetch the first streamed
  // fragments of the NEXT pass (the weights do not change during a launch), so a pass neither
  // ends on a wait for fragments nobody reads nor starts with an exposed prologue.
  f32x4 wb[PF][TO];
  typedef const __attribute__((address_space(1))) char* gchar_p;
  typedef const __attribute__((address_space(1))) f32x4* gf32x4_p;
  const unsigned lane_off16 = (unsigned)lane * 16u;
  const char* p16w = (const char*)pp.p16 + (size_t)wave * TO * NT * 256 * sizeof(float);
  auto issue = [&](int l, int ti, int stage) {
    // uniform (SGPR) base + one per-lane offset register
    const char* lb = p16w + (size_t)l * Hp * Hp * sizeof(float);
    if (NT > SWEEP_SCALAR_NT) asm volatile("" : "+s"(lb));   // keep it scalar: per-item 64-bit lane addresses would be hoisted and spill
#pragma unroll
    for (int to = 0; to < TO; ++to) {
      gchar_p base = (gchar_p)lb + (size_t)(to * NT + ti) * 256 * sizeof(float);
      wb[stage][to] = *(gf32x4_p)(base + lane_off16);
    }
  };
  const int l_last = n_hidden - 1;
  // first stream item of a pass: layer 0's first streamed k-tile, or (all of layer 0 resident)
  // layer 1's first k-tile
  const int ring_l0 = RT < NT ? 0 : min(1, l_last);
  constexpr int ring_t0 = RT < NT ? RT : 0;
  if (!SW && n_hidden > 0) {
#pragma unroll
    for (int st = 0; st < PF - 1; ++st) issue(ring_l0, ring_t0 + st, st);
  }
  // split sampler: two-item ring; an item = (layer, k-step) x both tiles x three terms.  Stream order per step:
  // layer 0 k-steps RES .. 7, then every k-step of the later layers; the last item of a step fetches the first of
  // the next (the weights do not change during a launch).  Both counts are even, so an item's slot is kt & 1.
  static_assert(!SW || ((8 - RES) % 2 == 0 && RES < 8), "ring slots continue across layers only for an even item count");
  SwFrag ring_s[2][TO];
  auto sw_issue = [&](int l, int kt, int slot) {
#pragma unroll
    for (int to = 0; to < TO; ++to) ring_s[slot][to] = sw_frag(l, kt, to);
  };
  if (SW && n_hidden > 0) sw_issue(0, RES, RES & 1);
  auto forward = [&](unsigned long long next_step) {
    // Every issue below is unconditional so that the compiler can count vmcnt exactly; a load
    // issued under a runtime condition makes it wait for ALL outstanding loads at the next use.
    int cur = 0;
    // FS = first streamed k-tile of the layer (RT for layer 0, 0 afterwards)
    auto layer = [&](int l, auto fs_c, auto first_c, auto draw_c) {
      constexpr int FS = decltype(fs_c)::value;
      // layer 0 carries the Philox pieces: in its resident k-tiles, or (no resident tile) in all.
      // DRAW: this wave owns chains (waves 0-3 of an 8-wave workgroup); with SWEEP_SPLIT the others
      // run a copy of the layer without the pieces -- their draws would be for chains that do not
      // exist, and VALU instructions between MFMAs are not free
      constexpr bool DRAW = decltype(draw_c)::value;
      constexpr bool FIRST = decltype(first_c)::value && DRAW;
      constexpr int PT = FIRST ? (FS > 0 ? FS : NT) : 0;
      SWEEP_STAMP(FS > 0 ? 7 : 11)
      __syncthreads();
      SWEEP_STAMP(FS > 0 ? 8 : 12)
      const f32x4* xin = (const f32x4*)(s_x + cur * NT * 256) + lane;
      f32x4 acc[TO];
#pragma unroll
      for (int to = 0; to < TO; ++to)
        acc[to] = *(const f32x4*)(s_bias + l * Hp + 16 * (wave * TO + to) + 4 * g);
      // resident k-tiles: operands straight from registers.  The 20 Philox rounds of the NEXT
      // step's two draws are cut into 2*FS pieces and pinned (sched_barrier) between groups of
      // 2*TO MFMAs, so the VALU work issues in the shadow of the matrix pipe.
      f32x4 inb[2];
      inb[0] = xin[0];
      uint4 cs[ND];
      uint2 ks[ND];
#pragma unroll
      for (int d = 0; d < ND; ++d) { cs[d] = ctr_of(d, next_step); ks[d] = key; }
      constexpr int NPIECE = 2 * PT, NROUND = 10 * ND;
      constexpr int RPP = NPIECE > 1 ? (NROUND + NPIECE - 2) / (NPIECE - 1) : NROUND;   // rounds / piece
      auto piece = [&](int p) {
#pragma unroll
        for (int q = p * RPP; q < (p + 1) * RPP && q < NROUND; ++q) {
#pragma unroll
          for (int d = 0; d < ND; ++d)
            if (q % ND == d) philox_round(cs[d], ks[d]);
        }
      };
#pragma unroll
      for (int ti = 0; ti < FS; ++ti) {
        if (ti + 1 < NT) inb[(ti + 1) & 1] = xin[(ti + 1) * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int to = 0; to < TO; ++to)
            acc[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(wres[ti * TO + to][r], inb[ti & 1][r],
                                                           acc[to], 0, 0, 0);
          if (DRAW && r == 1) { piece(2 * ti); __builtin_amdgcn_sched_barrier(0); }
          if (DRAW && r == 3) { piece(2 * ti + 1); __builtin_amdgcn_sched_barrier(0); }
        }
      }
      // the draws are only read by the NEXT step's proposals: without a use here LLVM sinks half
      // of the rounds below the layer, into the serial part of the step (one whole draw of the
      // two at config 3)
      auto pin_draw = [&]() {
        if (!SWEEP_PIN(UPRE)) return;
#pragma unroll
        for (int d = 0; d < ND; ++d)
          asm volatile("" : "+v"(cs[d].x), "+v"(cs[d].y), "+v"(cs[d].z), "+v"(cs[d].w));
      };
      if (DRAW && FS > 0) { pin_draw(); finish_draw(cs, next_step); }
      if (FS > 0) { SWEEP_STAMP(9) }
      // streamed k-tiles: weights PF-1 tiles ahead, activations one tile ahead
      if (FS == 0) inb[0] = xin[0];
#pragma unroll
      for (int ti = FS; ti < NT; ++ti) {
        const int tn = ti + PF - 1;
        if (tn < NT) issue(l, tn, (tn - FS) % PF);
        else issue(l < l_last ? l + 1 : ring_l0, l < l_last ? tn - NT : ring_t0 + tn - NT, (tn - FS) % PF);
        if (ti + 1 < NT) inb[(ti + 1) & 1] = xin[(ti + 1) * 64];
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetches ahead of this tile's MFMAs
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int to = 0; to < TO; ++to)
            acc[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[(ti - FS) % PF][to][r],
                                                           inb[ti & 1][r], acc[to], 0, 0, 0);
          if (FIRST && FS == 0) {
            if (r == 1) { piece(2 * ti); __builtin_amdgcn_sched_barrier(0); }
            if (r == 3) { piece(2 * ti + 1); __builtin_amdgcn_sched_barrier(0); }
          }
        }
      }
      if (FIRST && FS == 0) { pin_draw(); finish_draw(cs, next_step); }
      SWEEP_STAMP(FS > 0 ? 10 : 13)
      if (ACT == VMC_ACT_RELU_) vmc_mfma_settle_all(acc);   // the asm relu reads the accumulators (common.hpp)
      float* xout = s_x + (cur ^ 1) * NT * 256;
#pragma unroll
      for (int to = 0; to < TO; ++to) {
        finish_own(to, acc[to], l + 1 == n_hidden);
        if (l + 1 < n_hidden) *(f32x4*)(xout + ((wave * TO + to) * 64 + lane) * 4) = own[to];
      }
      if (save_acts) save_own(l + 1);
      cur ^= 1;
      SWEEP_STAMP(FS > 0 ? 11 : 14)
    };
    // split sampler: one layer = eight 32-deep k-steps; B operands (hi, mid, lo) from LDS one k-step ahead
    auto layer_s = [&](int l, auto first_c) {
      constexpr bool FIRSTL = decltype(first_c)::value;
      asm volatile("" : "+s"(sw_base));
      __syncthreads();
      const sw_u32x4* xin = (const sw_u32x4*)(s_x + cur * XB) + lane;       // [k-step][term][64 lanes]
      f32x4 acc[TO];
#pragma unroll
      for (int to = 0; to < TO; ++to)
        acc[to] = *(const f32x4*)(s_bias + l * Hp + 16 * (wave * TO + to) + 4 * g);
      sw_u32x4 xb[2][3];
#pragma unroll
      for (int t = 0; t < 3; ++t) xb[0][t] = xin[t * 64];
#pragma unroll
      for (int kt = 0; kt < 8; ++kt) {
        const bool resident = FIRSTL && kt < RES;
        if (!resident) {          // fetch the item behind this one (all issues unconditional: exact vmcnt)
          if (kt + 1 < 8) sw_issue(l, kt + 1, (kt + 1) & 1);
          else if (l < l_last) sw_issue(l + 1, 0, 0);
          else sw_issue(0, RES, RES & 1);
        }
        if (kt + 1 < 8) {
#pragma unroll
          for (int t = 0; t < 3; ++t) xb[(kt + 1) & 1][t] = xin[((kt + 1) * 3 + t) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetches ahead of this k-step's MFMAs
        const sw_u32x4 &xh = xb[kt & 1][0], &xm = xb[kt & 1][1], &xl = xb[kt & 1][2];
        // smallest terms first (the order of k_tail16s / k_tail16r)
#define SW_PRODUCT(WT, XT)                                                                                   \
        _Pragma("unroll") for (int to = 0; to < TO; ++to)                                                     \
          acc[to] = sw_mfma(resident ? wres_s[kt < RES ? kt : 0][to].WT : ring_s[kt & 1][to].WT, XT, acc[to]);
        SW_PRODUCT(l, xh) SW_PRODUCT(h, xl) SW_PRODUCT(m, xm) SW_PRODUCT(m, xh) SW_PRODUCT(h, xm) SW_PRODUCT(h, xh)
#undef SW_PRODUCT
      }
      vmc_mfma_settle_all(acc);                  // the relu reads the accumulators through inline asm (common.hpp)
#pragma unroll
      for (int to = 0; to < TO; ++to) finish_own(to, acc[to], l + 1 == n_hidden);
      if (l + 1 < n_hidden) sw_publish(s_x + (cur ^ 1) * XB);
      if (save_acts) save_own(l + 1);
      cur ^= 1;
    };
    if constexpr (SW) {
      layer_s(0, std::true_type{});
      for (int l = 1; l < n_hidden; ++l) layer_s(l, std::false_type{});
      return;
    }
    if ((HANDOFF_T && handoff) || (SWEEP_SPLIT(UPRE) && W1L && RT > 0 && NW > 4 && wave >= 4)) layer(0, std::integral_constant<int, RT>{}, std::true_type{}, std::false_type{});
    else layer(0, std::integral_constant<int, RT>{}, std::true_type{}, std::true_type{});
    for (int l = 1; l < n_hidden; ++l) layer(l, std::integral_constant<int, 0>{}, std::false_type{}, std::false_type{});
  };

  // output dot of the last activations (own) -> per-wave partial logits in s_part
  auto output_dot = [&]() {
    float part = 0.f;
    f32x4 w[TO];
#pragma unroll
    for (int to = 0; to < TO; ++to) w[to] = *(const f32x4*)(s_wout + 16 * (wave * TO + to) + 4 * g);
#pragma unroll
    for (int to = 0; to < TO; ++to)
#pragma unroll
      for (int e = 0; e < 4; ++e) part = fmaf(own[to][e], w[to][e], part);
    // sum over the four 16-lane rows: lane ^ 16, then lane ^ 32 (gfx950 row / half swaps: pure
    // VALU, the same two additions per lane as __shfl_xor without its two LDS-crossbar round trips)
    {
      const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(part), __float_as_uint(part), false, false);
      part = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
      const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(part), __float_as_uint(part), false, false);
      part = __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
    }
    if (g == 0) s_part[wave * 16 + j] = part;   // NW partials per chain
    __syncthreads();
  };

  auto logit_of = [&](int c) {   // fixed summation order over the NW per-wave partials
    float t = (s_part[c] + s_part[16 + c]) + (s_part[32 + c] + s_part[48 + c]);
    if (NW == 8) t += (s_part[64 + c] + s_part[80 + c]) + (s_part[96 + c] + s_part[112 + c]);
    return t + bout;
  };

  // it = -1: cache of the initial spins; 0..n_steps-1: mc_steps; n_steps: exact cache of the
  // final spins (all three share one instance of build/forward).
  // The outcome of iteration it-1 is resolved at the TOP of iteration it by the wave that owns
  // the chain (chains 4w..4w+3 -> wave w, all 16 lanes of a group redundantly, lane j == 0
  // writes): that wave is the only reader of the chain's spins in `proposals`, so no barrier
  // is needed between the Metropolis accept and the next proposal.
  unsigned int n_acc = 0;
  int prev_kind = 0;   // 0 none, 1 refresh, 2 step
  auto resolve = [&]() {
    if (prev_kind == 0 || (NW > 4 && wave >= 4)) return;
    const int c = my_c, gc = chain0 + c;
    // every LDS operand of the accept test is requested before the first one is used: one LDS
    // round trip for the phase instead of four (the asm pins the loads above the branches)
    float lg = s_logit[c], hl = s_hlu[c];
    int idn = s_idn[c], iup = s_iup[c];
    float ln = logit_of(c);
    asm volatile("" : "+v"(lg), "+v"(hl), "+v"(idn), "+v"(iup));
    float on_new = 0.f;
    if (RBM) {   // onsite term of the evaluated configuration: committed value (+ exchange update)
      on_new = s_on[c];
      if (prev_kind == 2) on_new += 2.f * (won_at(idn) - won_at(iup));
      ln += on_new;
    }
    if (prev_kind == 2) {
      // Metropolis accept (graph_builders.py:75-88)
      // exp(dlogit) > sqrt(u)  <=>  dlogit > 0.5 log(u)  (monotone; u = 0 always accepts)
      bool acc;
      if (a.oact == VMC_ACT_EXP_) acc = (ln - lg) > hl;
      else acc = vmc_out_accept(a.oact, ln, lg, s_u[c], hl);
      acc = acc && (gc < a.B);
      if (j == 0) {
        if (acc) {
          s_logit[c] = ln;
          if (RBM) s_on[c] = on_new;
          s_spin[c * Nst + idn] = 1.f;
          s_spin[c * Nst + iup] = -1.f;
          if (!W1L) s_sel[c] ^= 1;
          ++n_acc;
        }
        if (W1L) s_pacc[c] = acc ? 1 : 0;
        if (!FAST && a.acc_mask && gc < a.B) a.acc_mask[gc] = acc ? 1 : 0;
      }
    } else if (j == 0) {
      s_logit[c] = ln;
    }
  };
  long long it_first = -1;
  if (a.n_steps == 0) it_first = 0;   // a pure cache refresh (refresh_cache_by_sampler, vmc_api.hip): one pass, the final one
  if (a.cache_in_valid && a.n_steps > 0) {
    // the previous launch left an exact z1 / logit cache for these very chains: load it
    // instead of recomputing it (saves one of the two refresh passes per launch)
    for (int i = tid; i < 16 * (Hp / 4); i += NTH) {   // 16-byte loads, <= 2 per thread
      const int c = i / (Hp / 4), c4 = i % (Hp / 4), gc = chain0 + c;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (gc < a.B) v = *(const f32x4*)(a.z1_in + (long long)gc * Hp + 4 * c4);
      *(f32x4*)(s_z1 + c * ZS + 4 * c4) = v;
    }
    if (tid < 16) s_logit[tid] = chain0 + tid < a.B ? a.logit_in[chain0 + tid] : 0.f;
    onsite_direct();
    if (use_pref) draw_all(a.step0);
    if (HANDOFF_T && handoff) to_keys();
    __syncthreads();
    it_first = 0;
  }
  for (long long it = it_first; it <= a.n_steps; ++it) {
    const bool is_step = it >= 0 && it < a.n_steps;
    if (HANDOFF_T && handoff && wave >= 4) {
      draw_all(a.step0 + (unsigned long long)(it + 1));
      to_keys();
      float* dst = s_uh + ((wave - 4) * (UPRE + 1) * 64 + lane) * 4;
#pragma unroll
      for (int b = 0; b < UPRE; ++b)
        *(f32x4*)(dst + 256 * b) = f32x4{u_pre[4 * b], u_pre[4 * b + 1], u_pre[4 * b + 2], u_pre[4 * b + 3]};
      dst[256 * UPRE] = u_pre_acc;
    }
    save_acts = (it == a.n_steps) && (a.act_out != nullptr);
    stamp_on = is_step;
    if (STAMP) t0 = vmc_stamp();
    resolve();
    SWEEP_STAMP(5)
    if (is_step) {
      proposals(a.step0 + (unsigned long long)it);
    } else {
      __syncthreads();   // z1_direct reads every chain's (possibly just updated) spins
      if (tid < 16) { s_sel[tid] = 0; s_pacc[tid] = 0; }
      z1_direct();
      onsite_direct();
    }
    SWEEP_STAMP(0)
    __syncthreads();
    SWEEP_STAMP(1)
    if (HANDOFF_T && handoff && wave < 4) {   // the reads complete under build (next barrier at the latest)
      const float* src = s_uh + (wave * (UPRE + 1) * 64 + lane) * 4;
#pragma unroll
      for (int b = 0; b < UPRE; ++b) {
        const f32x4 ub = *(const f32x4*)(src + 256 * b);
#pragma unroll
        for (int e = 0; e < 4; ++e) u_pre[4 * b + e] = ub[e];
      }
      u_pre_acc = src[256 * UPRE];
    }
    build(is_step);
    SWEEP_STAMP(2)
    const unsigned long long next_step = a.step0 + (unsigned long long)(it + 1);
    // (the split layers carry no Philox pieces: without the hand-over the draws follow the layers un-overlapped)
    const bool pre_here = use_pref && n_hidden > 0 && !(SW && !(HANDOFF_T && handoff));
    if (n_hidden > 0) forward(next_step);
    if (use_pref && !pre_here && !(NW > 4 && wave >= 4)) draw_all(next_step);
    SWEEP_STAMP(3)
    output_dot();
    SWEEP_STAMP(4)
    prev_kind = is_step ? 2 : 1;
  }
  stamp_on = false;
  resolve();          // logit of the final refresh
  __syncthreads();
#undef SWEEP_STAMP
  if (STAMP && a.dbg_cycles && lane == 0) {
#pragma unroll
    for (int k = 0; k < 16; ++k) a.dbg_cycles[((long long)blockIdx.x * NW + wave) * 16 + k] = cyc[k];
  }

  // write back chains and the exact cache
  if (tid < 16 && chain0 + tid < a.B) a.logit[chain0 + tid] = s_logit[tid];
  if (RBM && tid < 16 && chain0 + tid < a.B) a.onsite[chain0 + tid] = s_on[tid];
  for (int i = tid; i < 16 * N; i += NTH) {
    const int c = i / N, n = i % N, gc = chain0 + c;
    if (gc < a.B) a.configs[(long long)gc * N + n] = s_spin[c * Nst + n];
  }
  for (int i = tid; i < 16 * (Hp / 4); i += NTH) {
    const int c = i / (Hp / 4), c4 = i % (Hp / 4), gc = chain0 + c;
    if (gc < a.B) *(f32x4*)(a.z1 + (long long)gc * Hp + 4 * c4) = *(const f32x4*)(s_z1 + c * ZS + 4 * c4);
  }
  if (j == 0 && n_acc) atomicAdd(a.accepted, (unsigned long long)n_acc);   // waves 0-3 only
  // Bond census of the chains this workgroup leaves behind (k_bond_count, eloc.hip: antiparallel bonds and
  // diag = sum 0.25 jz s_i s_j per chain, the same per-lane order of fused multiply-adds and the same xor
  // tree, so the local energies do not depend on who counted): the spins are still in LDS.  A wave takes
  // its chains one after the other, four 64-bond groups per pass with their bond loads issued together.
  if (a.cnt_out) {
    for (int c = wave; c < 16 && chain0 + c < a.B; c += NW) {
      const float* x = s_spin + c * Nst;
      float d = 0.f;
      int n = 0;
      for (int k0 = 0; k0 < a.n_bonds; k0 += 256) {
        int2 ab[4]; float q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = k0 + 64 * u + lane, kk = k < a.n_bonds ? k : 0;
          ab[u] = a.bonds[kk]; q[u] = a.quarter_jz[kk];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool in = k0 + 64 * u + lane < a.n_bonds;
          const float sz = x[ab[u].x] * x[ab[u].y];
          if (in) d = fmaf(q[u], sz, d);
          n += __popcll(__ballot(in && sz < 0.f));
        }
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) d += __shfl_xor(d, m);
      if (lane == 0) { a.cnt_out[chain0 + c] = n; a.diag_out[chain0 + c] = d; }
    }
  }
}

template <int NT, int NW, int RTP, bool STAMP, bool W1L, bool FAST, int UPRE, bool RBM, int ACT>
__global__ __launch_bounds__(NW * 64, NT == 16 ? (RTP == 0 ? SWEEP_OCC0 : 2) : 1) void k_sweep16(SweepArgs a) {
  sweep16_body<NT, NW, RTP, STAMP, W1L, FAST, UPRE, RBM, ACT>(a);
}
// the 3 x bf16 split sampler (instantiated by sweep_split.hip only)
template <int RTP, bool FAST, int UPRE = 2>
__global__ __launch_bounds__(512, 2) void k_sweep16s(SweepArgs a) {
  sweep16_body<16, 8, RTP, false, false, FAST, UPRE, false, VMC_ACT_RELU_, true>(a);
}

// (LDS bytes and the choice of variant: plan_sweep_lds_bytes / plan_sweep, plan.hpp)
template <int NT, int NW, int RTP, bool RBM, int ACT>
static hipError_t launch_sweep16_t(hipStream_t s, const SweepArgs& a_in) {
  SweepArgs a = a_in;
  const dim3 grid((a.B + 15) / 16), block(NW * 64);
  constexpr bool TUNED = ACT == VMC_ACT_RELU_;   // other activations only get the general variant
  const bool plain = a.inj_up == nullptr && a.dbg_up == nullptr;
  const SweepPlan sp = plan_sweep(a.N, NT, NW, a.n_hidden, RBM, a.no_w1l != 0, plain, TUNED);
  if (!sp.ok) return hipErrorInvalidValue;
  const bool w1l = sp.w1l != 0, fast2 = sp.fast == 2, fast4 = sp.fast == 2 || sp.fast == 4;
  const size_t lds = sp.lds;
  a.uh_lds = sp.uh_lds;
#define SWEEP_LAUNCH(ST, WL, FA, UP)                                                          \
  do {                                                                                        \
    hipError_t e = hipFuncSetAttribute((const void*)k_sweep16<NT, NW, RTP, ST, WL, FA, UP, RBM, ACT>, \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return e;                                                            \
    hipLaunchKernelGGL((k_sweep16<NT, NW, RTP, ST, WL, FA, UP, RBM, ACT>), grid, block, lds, s, a); \
    return hipGetLastError();                                                                 \
  } while (0)
  if (a.dbg_cycles) {
    if (!(w1l && fast2) || RBM || NT != 16 || !TUNED) return hipErrorInvalidValue;   // diagnostic build: production variant only
    if constexpr (!RBM && NT == 16 && TUNED) SWEEP_LAUNCH(true, true, true, 2);
  }
  if constexpr (TUNED) {
    if constexpr (NT <= 16) {
      if (w1l) {
        if (fast2) SWEEP_LAUNCH(false, true, true, 2);
        if (fast4) SWEEP_LAUNCH(false, true, true, 4);
      }
    }
    if (!w1l) {
      if (fast2) SWEEP_LAUNCH(false, false, true, 2);
      if (fast4) SWEEP_LAUNCH(false, false, true, 4);
    }
  }
  if constexpr (NT <= 16) {
    if (w1l) SWEEP_LAUNCH(false, true, false, 2);
  }
  SWEEP_LAUNCH(false, false, false, 2);
#undef SWEEP_LAUNCH
}

// H = 256 runs 8 waves (2 per SIMD, wave w owns output units 32w..32w+31) with SWEEP_RT resident
// k-tiles.  A 4-wave variant (4 output tiles per wave, two workgroups per CU at <= 256
// registers) was measured in round 2 and dropped: it spills 80-160 VGPRs at any number of
// resident tiles and ran 1.42 ms per sweep against 1.14 ms alone, 2.92 against 2.18 ms with two
// workgroups per CU (8192 chains), 7.1 against 6.1 ms at config 5.
template <bool RBM, int ACT>
static hipError_t launch_sweep16_r(hipStream_t s, const SweepArgs& a, int Hp) {
  switch (Hp / 16) {
#ifndef VMC_QUICK   // development builds (-DVMC_QUICK) only instantiate H = 256, fully_connected
    case 4: return launch_sweep16_t<4, 4, SWEEP_RT, RBM, ACT>(s, a);
    case 8: return launch_sweep16_t<8, 4, SWEEP_RT, RBM, ACT>(s, a);
    case 12: return launch_sweep16_t<12, 4, SWEEP_RT, RBM, ACT>(s, a);
#endif
    case 16: return launch_sweep16_t<16, 8, SWEEP_RT, RBM, ACT>(s, a);
#ifndef VMC_QUICK
    // 257 .. 512 units (384 / 512 padded): wave w owns 3 or 4 output tiles, W1 in L2, the weight
    // stream addressed from a scalar base (SWEEP_SCALAR_NT).  relu gets the prefetched-Philox
    // variants, the other activations the general one (as at <= 256 units); both dense ansatz types.
    case 24: return launch_sweep16_t<24, 8, SWEEP_RT_WIDE, RBM, ACT>(s, a);
    case 32: return launch_sweep16_t<32, 8, SWEEP_RT_WIDE, RBM, ACT>(s, a);
#endif
    default: return hipErrorInvalidValue;
  }
}

