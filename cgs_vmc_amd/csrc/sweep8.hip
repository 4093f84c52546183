// k_sweep8: the Metropolis exchange sampler (graph_builders.py:38-89) with EIGHT chains per workgroup, for batches that
// leave k_sweep16's sixteen-chain tiles on at most half of the chip (1,024 chains per GPU = 64 tiles on 256 CUs: BASELINE
// configs 2 and 5).  Same chains, bit for bit: every floating-point result a Metropolis decision
// depends on is produced by the same sequence of fused multiply-adds and additions as in k_sweep16
// (tests/test_gpu_sweep8.py compares chains, logits, z1 cache, activations and census of the two kernels).
//
// The H x H layers run on v_mfma_f32_4x4x1_16B_f32 (sixteen 4 x 4 blocks, k = 1: 512 flops in 2 passes, the matrix
// rate of the 16x16x4 form).  tools/ubench/mfma_order.hip established the two facts this rests on
// (profiles/r6_mfma_order_probe.txt):
//   * v_mfma_f32_16x16x4_f32 accumulates its four k slots as a chain of fused multiply-adds in slot order, so k = 1
//     MFMAs issued in k_sweep16's k order (tile ti, register r, slot g: unit 16 ti + 4 g + r) give its bits;
//   * the operand broadcasts: CBSZ = 3 / ABID = j hands block j of each group of eight blocks to all eight as the A
//     operand, BLGP = 1 / 2 hands lanes 0-31 / 32-63 of the B operand to both halves.
// Wave w owns output units 32 w .. 32 w + 31 (NW = Hp / 32 waves).  Per MFMA: A = activations x[k][chain], rows =
// chains 4 (lane / 32) + v; B = weights W[k][32 w + lane % 32]; D: lane <-> unit 32 w + lane % 32, register v <-> chain
// 4 (lane / 32) + v.  The two halves of a weight register hold two consecutive k of the order (BLGP picks one), so
// every loaded weight feeds TWO MFMAs (8 chains) and the fragments come straight out of k_sweep16's image p16 with
// another lane -> address map: no second parameter image.  One A register holds eight k (ABID picks one), read from
// LDS as 16 bytes per lane = 32 k per ds_read_b128: LDS traffic is 1/8 of an operand per MFMA.
//
// Chain c belongs to a group of LPC = 64 NW / 8 lanes (a whole wave at 256 units): lane i of the group owns site
// block i (4 sites: spins and the step's sortable uniform keys in registers), draws its Philox block, and the group
// resolves the Metropolis test and proposes the next exchange with lane reductions only.  The group also computes the
// output dot of its chain from the last activations in LDS, in k_sweep16's summation tree, so there is no partial-sum
// hand-over and the step has n_hidden + 2 barriers (as k_sweep16).
// Committed z1 lives in registers (lane <-> unit, 4 chains); an accepted move is folded in lazily at the next build.
// Weights: layer 0 (and R1 fragments of layer 1) stay in registers for the launch, the rest streams L2 -> registers
// through a PF-slot ring that runs across layers and steps.
// fully_connected + relu only (k_sweep16 serves the other activations and the RBM), plain launches only (no injected
// proposals / debug dump: those take k_sweep16, which produces the same chains).
#include "common.hpp"

#include <cstdlib>
#include <type_traits>

#ifndef SWEEP8_PF
#define SWEEP8_PF 4     // ring slots (fragments of 16 bytes per lane); even
#endif

namespace {

template <int CTRL>
__device__ __forceinline__ float d8_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ unsigned d8_u(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
#define D8_XOR1 0xB1
#define D8_XOR2 0x4E
#define D8_HALF_MIRROR 0x141
#define D8_MIRROR 0x140

__device__ __forceinline__ void settle1(f32x4& a) { asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a)); }

// position of unit u in the order the MFMAs consume k: within a 16-unit tile register r outer, slot g inner
__device__ __forceinline__ int m_of(int u) { return (u & ~15) | ((u & 3) << 2) | ((u >> 2) & 3); }

// s_memtime stamp for the diagnostic instantiation (STAMP = true) only
__device__ __forceinline__ unsigned long long stamp8() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

// STAMP (vmc_debug_sweep_profile): s_memtime ticks per mc_step and wave of the phases
//   0 resolve (output dot + accept)   1 proposals   2 barrier A   3 build (+ Philox draw)   4 layer 0: barrier
//   5 layer 0: MFMAs   6 layer 0: epilogue   7 later layers: barriers   8 later layers: MFMAs   9 later layers: epilogues
//   10 end barrier
template <int NW, int R0, int R1, bool W1L, bool STAMP>
__global__ __launch_bounds__(NW * 64) void k_sweep8(SweepArgs a) {
  constexpr int NTH = NW * 64, Hp = 32 * NW, NT = Hp / 16, NI = 2 * NT;   // NI fragments per layer and wave
  constexpr int S = Hp + 16, W1S = Hp + 4, LPC = 8 * NW, PF = SWEEP8_PF;
  static_assert(NW == 4 || NW == 8, "128 or 256 units");
  static_assert(R0 == NI, "layer 0 is register resident");
  static_assert(PF % 2 == 0 && NI % PF == 0 && (NI - R1) % PF == 0 && R1 % 2 == 0, "ring turns are whole per layer");
  constexpr int SL = R1 == NI ? 2 : 1;       // first layer with streamed fragments
  constexpr int FS_SL = SL == 1 ? R1 : 0;    // its first streamed fragment
  extern __shared__ float smem[];
  const int N = a.N, Nst = (N + 3) & ~3, nblk = Nst >> 2;
  float* s_spin = smem;                       // [8][Nst]
  float* s_x = s_spin + 8 * Nst;              // [2][8][S]
  int* s_iup = (int*)(s_x + 2 * 8 * S);       // [8]
  int* s_idn = s_iup + 8;                     // [8]
  int* s_pacc = s_idn + 8;                    // [8]
  float* s_wout = (float*)(s_pacc + 8);       // [Hp]
  float* s_bias = s_wout + Hp;                // [n_hidden][Hp]
  float* s_w1 = s_bias + a.n_hidden * Hp;     // [N][W1S] (W1L)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chain0 = blockIdx.x * 8;
  const PackedParams& pp = a.pp;
  const int n_hidden = a.n_hidden;
  // compute role: unit u, chains 4 hh + v
  const int ul = lane & 31, hh = lane >> 5, u = 32 * wave + ul, mu = m_of(u);
  // owner role: chain oc of the tile, site block blk
  const int oc = (tid / LPC), blk = tid % LPC, gc_own = chain0 + oc;
  const uint32_t my_gid = (uint32_t)(a.chain_offset + gc_own);
  const bool own_ok = gc_own < a.B;
  const uint2 key = make_uint2(a.seed_lo, a.seed_hi);
  const float bout = pp.bout[0];

  // ---- launch prologue: spins (registers of the owner + LDS), constants, W1
  f32x4 sp;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int n = 4 * blk + e;
    float v = 0.f;
    if (n < N) v = own_ok ? a.configs_in[(long long)gc_own * N + n] : ((n & 1) ? -1.f : 1.f);
    sp[e] = v;
  }
  auto spins_to_lds = [&]() { if (blk < nblk) *(f32x4*)(s_spin + oc * Nst + 4 * blk) = sp; };
  spins_to_lds();
  for (int i = tid; i < n_hidden * Hp; i += NTH) s_bias[i] = pp.bh[i];
  for (int i = tid; i < Hp; i += NTH) s_wout[i] = pp.woutp[i];
  if (tid < 8) { s_pacc[tid] = 0; s_iup[tid] = 0; s_idn[tid] = 0; }
  if (W1L) {
    const int total = N * (Hp / 4);
    for (int base = tid; base < total; base += 8 * NTH) {
      f32x4 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = min(base + q * NTH, total - 1);
        v[q] = *(const f32x4*)(pp.w1p + (long long)(i / (Hp / 4)) * Hp + 4 * (i % (Hp / 4)));
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = base + q * NTH;
        if (i < total) *(f32x4*)(s_w1 + (i / (Hp / 4)) * W1S + 4 * (i % (Hp / 4))) = v[q];
      }
    }
  }

  // ---- weight fragments.  Fragment q = 2 ti + p of layer l (this lane): p16[l][to = 2 wave + (ul >> 4)][ti][(2 p + hh) * 16 + (lane & 15)][0..3]
  //      = W_l[16 ti + 4 (2 p + hh) + r][u], r = 0..3
  typedef const __attribute__((address_space(1))) char* gchar_p;
  typedef const __attribute__((address_space(1))) f32x4* gf32x4_p;
  const char* p16w = (const char*)pp.p16 + (size_t)(2 * wave) * NT * 1024;
  const unsigned lane_off = (unsigned)(((lane >> 4) & 1) * NT * 1024 + ((lane >> 5) * 16 + (lane & 15)) * 16);
  // (scalar base + one 32-bit lane offset + an immediate: with a visible base the compiler hoists a 64-bit lane
  // address per fragment out of the step loop and spills them -- k_sweep16's issue() has the same cure)
  auto wload_at = [&](const char* lb, int q) -> f32x4 {
    return *(gf32x4_p)((gchar_p)lb + (size_t)((q >> 1) * 1024 + (q & 1) * 512) + lane_off);
  };
  auto layer_base = [&](int l) {
    const char* lb = p16w + (size_t)l * Hp * Hp * sizeof(float);
    asm volatile("" : "+s"(lb));
    return lb;
  };
  auto wload = [&](int l, int q) -> f32x4 { return wload_at(layer_base(l), q); };
  f32x4 wres0[R0];
  f32x4 wres1[R1 > 0 ? R1 : 1];
#pragma unroll
  for (int q = 0; q < R0; ++q) wres0[q] = wload(0, q);
  if (n_hidden > 1) {
#pragma unroll
    for (int q = 0; q < R1; ++q) wres1[q] = wload(1, q);
  }
  f32x4 ring[PF];
  const bool streaming = n_hidden > SL;
  if (streaming) {
#pragma unroll
    for (int st = 0; st < PF - 2; ++st) ring[st] = wload(SL, FS_SL + st);
  }
  __syncthreads();

  unsigned long long cyc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;
  bool stamp_on = false;
#define SW8_STAMP(k) \
  if (STAMP) { const unsigned long long t1_ = stamp8(); if (stamp_on) cyc[k] += t1_ - t0; t0 = t1_; }

  // ---- owner helpers
  unsigned ukey[4] = {0u, 0u, 0u, 0u};   // sortable keys of the NEXT proposal: (24-bit draw << 8) | (255 - site); 0 beyond the lattice
  float uacc_next = 0.f;
  // The uniforms of `step`: site block blk -> four keys; the acceptance uniform from block VMC_ACCEPT_BLOCK, word 0
  // (graph_builders.py:59, 76-77).  When the lattice leaves the group's last lane without a site block it draws the
  // acceptance block instead; otherwise every lane makes a second call.
  // (The draw runs in build(), behind the issue of the W1 row reads: at 256 sites those come from L2 and the ~350
  // clocks of a Philox call vanish in their latency.  Riding the rounds between layer 0's MFMA rows was measured and
  // dropped: both waves of a SIMD reach them together, layer 0 grew by more than build() shrank.)
  const bool spare = nblk < LPC;
  auto draw = [&](unsigned long long step) {
    const uint32_t b0 = (spare && blk == LPC - 1) ? VMC_ACCEPT_BLOCK : (uint32_t)blk;
    const uint4 r = philox4x32_10_wide(make_uint4(b0, my_gid, (uint32_t)step, (uint32_t)(step >> 32)), key);
    const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = 4 * blk + e;
      ukey[e] = n < N ? ((rr[e] >> 8) << 8) | (unsigned)(255 - n) : 0u;
    }
    if (spare) {
      uacc_next = __shfl(u32_to_uniform(r.x), lane | (LPC - 1));
    } else {
      const uint4 q = philox4x32_10_wide(make_uint4(VMC_ACCEPT_BLOCK, my_gid, (uint32_t)step, (uint32_t)(step >> 32)), key);
      uacc_next = u32_to_uniform(q.x);
    }
  };
  // max over the LPC lanes of a chain group
  auto group_max = [&](unsigned x) {
    x = max(x, d8_u<D8_XOR1>(x)); x = max(x, d8_u<D8_XOR2>(x));
    x = max(x, d8_u<D8_HALF_MIRROR>(x)); x = max(x, d8_u<D8_MIRROR>(x));
    { const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false); x = max(r[0], r[1]); }
    if (LPC == 64) { const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false); x = max(r[0], r[1]); }
    return x;
  };
  // output dot of chain oc from the last activations (plain unit order in s_x[n_hidden & 1]), k_sweep16's tree:
  // partial (W, g) = fma chain over units 32 W + 16 to + 4 g + e (to, e ascending), then g ^ 1, g ^ 2, then the waves
  // ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7)), then + b_out
  const float* s_last = s_x + (n_hidden & 1) * 8 * S + oc * S;
  const int dl = blk % (4 * NW), dW = dl >> 2, dg = dl & 3;
  f32x4 wo[2];
#pragma unroll
  for (int to = 0; to < 2; ++to) wo[to] = *(const f32x4*)(s_wout + 32 * dW + 16 * to + 4 * dg);
  auto chain_logit = [&]() {
    float part = 0.f;
    f32x4 x[2];
#pragma unroll
    for (int to = 0; to < 2; ++to) x[to] = *(const f32x4*)(s_last + 32 * dW + 16 * to + 4 * dg);
#pragma unroll
    for (int to = 0; to < 2; ++to)
#pragma unroll
      for (int e = 0; e < 4; ++e) part = fmaf(x[to][e], wo[to][e], part);
    part += d8_f<D8_XOR1>(part);
    part += d8_f<D8_XOR2>(part);
    part += d8_f<D8_HALF_MIRROR>(part);
    part += d8_f<D8_MIRROR>(part);
    if (NW == 8) {
      const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(part), __float_as_uint(part), false, false);
      part = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    return part + bout;
  };

  // ---- compute-role state
  f32x4 zreg = {0.f, 0.f, 0.f, 0.f};    // committed z1 of unit u, chains 4 hh + v
  f32x4 dprev = {0.f, 0.f, 0.f, 0.f};   // W1[i_dn] - W1[i_up] of the previous proposal
  f32x4 own;                            // activations of the stage in flight
  bool save_acts = false;
  auto save_own = [&](int l) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int gc = chain0 + 4 * hh + v;
      if (gc < a.B) a.act_out[((long long)l * a.B + gc) * Hp + u] = own[v];
    }
  };
  // exact first layer of the spins in LDS: b1 + sum_n s_n W1[n][u], n ascending (k_sweep16's z1_direct)
  auto z1_direct = [&]() {
    f32x4 acc;
    const float b = pp.b1p[u];
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[v] = b;
    const float* sp0 = s_spin + (4 * hh) * Nst;
    if (W1L) {
      for (int n = 0; n < N; ++n) {
        const float w = s_w1[n * W1S + u];
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[v] = fmaf(sp0[v * Nst + n], w, acc[v]);
      }
    } else {
      int n = 0;
      for (; n + 8 <= N; n += 8) {
        float w[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) w[q] = pp.w1p[(long long)(n + q) * Hp + u];
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[v] = fmaf(sp0[v * Nst + n + q], w[q], acc[v]);
      }
      for (; n < N; ++n) {
        const float w = pp.w1p[(long long)n * Hp + u];
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[v] = fmaf(sp0[v * Nst + n], w, acc[v]);
      }
    }
    zreg = acc;
  };
  // candidate z1 and the first activations -> operand buffer 0 (k order), or plain order when there is no H x H layer
  auto build = [&](bool with_delta, unsigned long long draw_step, bool do_draw) {
    f32x4 zc = zreg;
    if (with_delta) {
      const int4 idn = *(const int4*)(s_idn + 4 * hh), iup = *(const int4*)(s_iup + 4 * hh), pa = *(const int4*)(s_pacc + 4 * hh);
      const int dn[4] = {idn.x, idn.y, idn.z, idn.w}, up[4] = {iup.x, iup.y, iup.z, iup.w}, pc[4] = {pa.x, pa.y, pa.z, pa.w};
      float x1[4], y1[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        if (W1L) { x1[v] = s_w1[dn[v] * W1S + u]; y1[v] = s_w1[up[v] * W1S + u]; }
        else { x1[v] = pp.w1p[(long long)dn[v] * Hp + u]; y1[v] = pp.w1p[(long long)up[v] * Hp + u]; }
      }
      if (do_draw) draw(draw_step);     // the Philox rounds run while the row reads are in flight
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        zreg[v] = fmaf(pc[v] != 0 ? 2.f : 0.f, dprev[v], zreg[v]);   // the previous step's accepted move
        dprev[v] = x1[v] - y1[v];
        zc[v] = fmaf(2.f, dprev[v], zreg[v]);                        // candidate z1'
      }
    } else if (do_draw) {
      draw(draw_step);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) own[v] = vmc_act<VMC_ACT_RELU_>(zc[v]);
    float* dst = s_x + (4 * hh) * S + mu;
#pragma unroll
    for (int v = 0; v < 4; ++v) dst[v * S] = own[v];
    if (save_acts) save_own(0);
  };

  // ---- H x H layers
  // the four MFMAs of register r of a tile pair (slots g = 0..3 ascending: k_sweep16's order); JB = 4 (ti & 1)
#define SW8_MFMA(R, JB, XE, W, BL) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(xa_[XE], W[R], acc, 3, JB + R, BL);
#define SW8_ROW(R, JB) SW8_MFMA(R, JB, 0, w0, 1) SW8_MFMA(R, JB, 1, w0, 2) SW8_MFMA(R, JB, 2, w1, 1) SW8_MFMA(R, JB, 3, w1, 2)
  const int l_last = n_hidden - 1;
  // FL: 0 = layer 0 (all fragments resident), 1 = layer 1 (R1 resident, the rest streamed), 2 = streamed
  auto layer = [&](int l, auto fl_c) {
    constexpr int FL = decltype(fl_c)::value;
    constexpr int FS = FL == 0 ? NI : (FL == 1 ? R1 : 0);    // first streamed fragment
    __syncthreads();
    SW8_STAMP(FL == 0 ? 4 : 7)
    const float* xrow = s_x + (l & 1) * 8 * S + (4 * hh + (lane & 3)) * S + 4 * ((lane >> 2) & 7);
    f32x4 acc;
    {
      const float b = s_bias[l * Hp + u];
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[v] = b;
    }
    // streamed fragments beyond this layer's last belong to the next streaming layer, or to the first one of the
    // next step (the weights do not change during a launch); its first streamed fragment is folded into the base
    const char* lb_cur = nullptr;
    const char* lb_next = nullptr;
    if (FS < NI) {
      const bool wrap = l >= l_last;
      const int ln = wrap ? SL : l + 1;
      lb_cur = layer_base(l);
      lb_next = layer_base(ln) + (ln == 1 ? R1 * 512 : 0);
    }
    f32x4 xa[2];
    xa[0] = *(const f32x4*)xrow;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
      if ((ti & 1) == 0 && ti + 2 < NT) xa[((ti >> 1) + 1) & 1] = *(const f32x4*)(xrow + 32 * ((ti >> 1) + 1));
      f32x4 w0, w1;
      if (2 * ti < FS) {
        w0 = FL == 0 ? wres0[2 * ti] : wres1[FL == 1 ? 2 * ti : 0];
        w1 = FL == 0 ? wres0[2 * ti + 1] : wres1[FL == 1 ? 2 * ti + 1 : 0];
      } else {
        // two fragments PF - 2 ahead of this tile's pair; every issue unconditional (exact vmcnt)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int qn = 2 * ti + p + PF - 2;
          ring[(qn - FS) % PF] = qn < NI ? wload_at(lb_cur, qn) : wload_at(lb_next, qn - NI);
        }
        __builtin_amdgcn_sched_barrier(0);
        w0 = ring[(2 * ti - FS) % PF];
        w1 = ring[(2 * ti + 1 - FS) % PF];
      }
      const f32x4& xa_ = xa[(ti >> 1) & 1];
      if (ti & 1) { SW8_ROW(0, 4) SW8_ROW(1, 4) SW8_ROW(2, 4) SW8_ROW(3, 4) }
      else { SW8_ROW(0, 0) SW8_ROW(1, 0) SW8_ROW(2, 0) SW8_ROW(3, 0) }
    }
    settle1(acc);
    SW8_STAMP(FL == 0 ? 5 : 8)
#pragma unroll
    for (int v = 0; v < 4; ++v) own[v] = vmc_act<VMC_ACT_RELU_>(acc[v]);
    float* dst = s_x + ((l + 1) & 1) * 8 * S + (4 * hh) * S + (l + 1 < n_hidden ? mu : u);
#pragma unroll
    for (int v = 0; v < 4; ++v) dst[v * S] = own[v];
    if (save_acts) save_own(l + 1);
    SW8_STAMP(FL == 0 ? 6 : 9)
  };

  // ---- the step loop.  it = -1: exact cache of the initial spins; 0 .. n_steps - 1: mc_steps; n_steps: exact cache
  // of the final spins (k_sweep16's schedule).  The outcome of iteration it - 1 is resolved at the top of iteration it.
  float logit_c = 0.f;        // committed logit of chain oc
  float hl_cur = 0.f, u_cur = 0.f;
  int iup_cur = 0, idn_cur = 0;
  unsigned n_acc = 0;
  int prev_kind = 0;          // 0 none, 1 refresh, 2 step
  long long it_first = -1;
  if (a.n_steps == 0) it_first = 0;
  if (a.cache_in_valid && a.n_steps > 0) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int gc = chain0 + 4 * hh + v;
      zreg[v] = gc < a.B ? a.z1_in[(long long)gc * Hp + u] : 0.f;
    }
    logit_c = own_ok ? a.logit_in[gc_own] : 0.f;
    it_first = 0;
  }
  if (a.n_steps > 0 && it_first == 0) draw(a.step0);
  for (long long it = it_first; it <= a.n_steps; ++it) {
    const bool is_step = it >= 0 && it < a.n_steps;
    save_acts = (it == a.n_steps) && (a.act_out != nullptr);
    stamp_on = is_step;
    if (STAMP) t0 = stamp8();
    // resolve the previous iteration (the owner group of the chain; every lane of the group holds the same values)
    if (prev_kind != 0) {
      const float ln = chain_logit();
      if (prev_kind == 2) {
        // Metropolis accept (graph_builders.py:75-88): exp(dlogit) > sqrt(u)  <=>  dlogit > 0.5 log u
        bool acc;
        if (a.oact == VMC_ACT_EXP_) acc = (ln - logit_c) > hl_cur;
        else acc = vmc_out_accept(a.oact, ln, logit_c, u_cur, hl_cur);
        acc = acc && own_ok;
        if (acc) {
          logit_c = ln;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int n = 4 * blk + e;
            sp[e] = n == idn_cur ? 1.f : (n == iup_cur ? -1.f : sp[e]);
          }
          if (blk == 0) ++n_acc;
        }
        if (blk == 0) s_pacc[oc] = acc ? 1 : 0;
      } else {
        logit_c = ln;
      }
    }
    SW8_STAMP(0)
    if (is_step) {
      // proposals (graph_builders.py:59-65): argmax / argmin of s * u with the first-index tie rule = two integer
      // max reductions over the sortable keys of the up / down spins (k_sweep16's hand-over variants)
      unsigned kup = 0u, kdn = 0u;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        kup = max(kup, sp[e] > 0.f ? ukey[e] : 0u);
        kdn = max(kdn, sp[e] < 0.f ? ukey[e] : 0u);
      }
      kup = group_max(kup); kdn = group_max(kdn);
      // a chain without an up (or a down) spin has no exchange move: null move 0 <-> 0 with a NaN acceptance uniform,
      // which every accept test rejects (k_sweep16, same convention)
      const bool none = (kup == 0u) | (kdn == 0u);
      u_cur = none ? __uint_as_float(0x7fc00000u) : uacc_next;
      hl_cur = 0.5f * __logf(u_cur);
      iup_cur = none ? 0 : 255 - (int)(kup & 255u);   // argmax of s*u: the UP spin to lower   (graph_builders.py:64-65)
      idn_cur = none ? 0 : 255 - (int)(kdn & 255u);   // argmin of s*u: the DOWN spin to raise (graph_builders.py:62-63)
      if (blk == 0) { s_iup[oc] = iup_cur; s_idn[oc] = idn_cur; }
    } else {
      spins_to_lds();
      if (blk == 0) s_pacc[oc] = 0;
    }
    SW8_STAMP(1)
    __syncthreads();
    SW8_STAMP(2)
    if (!is_step) { z1_direct(); dprev = f32x4{0.f, 0.f, 0.f, 0.f}; }
    build(is_step, a.step0 + (unsigned long long)(it + 1), it + 1 < a.n_steps);
    SW8_STAMP(3)
    {
      typedef std::integral_constant<int, 0> c0; typedef std::integral_constant<int, 1> c1; typedef std::integral_constant<int, 2> c2;
      layer(0, c0{});
      if (n_hidden > 1) {
        if (R1 > 0) layer(1, c1{});
        else layer(1, c2{});
      }
      for (int l = 2; l < n_hidden; ++l) layer(l, c2{});
    }
    __syncthreads();
    SW8_STAMP(10)
    prev_kind = is_step ? 2 : 1;
  }
#undef SW8_STAMP
  logit_c = chain_logit();    // logit of the final refresh
  if (STAMP && a.dbg_cycles && lane == 0) {
#pragma unroll
    for (int k = 0; k < 16; ++k) a.dbg_cycles[((long long)blockIdx.x * NW + wave) * 16 + k] = cyc[k];
  }

  // ---- write back chains and the exact cache
  if (blk == 0 && own_ok) a.logit[gc_own] = logit_c;
  if (own_ok) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = 4 * blk + e;
      if (n < N) a.configs[(long long)gc_own * N + n] = sp[e];
    }
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int gc = chain0 + 4 * hh + v;
    if (gc < a.B) a.z1[(long long)gc * Hp + u] = zreg[v];
  }
  if (blk == 0 && n_acc) atomicAdd(a.accepted, (unsigned long long)n_acc);
  // bond census of the chains this workgroup leaves behind (k_bond_count's arithmetic, as k_sweep16): the final
  // refresh left the spins in LDS
  if (a.cnt_out) {
    for (int c = wave; c < 8 && chain0 + c < a.B; c += NW) {
      const float* x = s_spin + c * Nst;
      float d = 0.f;
      int n = 0;
      for (int k0 = 0; k0 < a.n_bonds; k0 += 256) {
        int2 ab[4]; float q[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const int k = k0 + 64 * w + lane, kk = k < a.n_bonds ? k : 0;
          ab[w] = a.bonds[kk]; q[w] = a.quarter_jz[kk];
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const bool in = k0 + 64 * w + lane < a.n_bonds;
          const float sz = x[ab[w].x] * x[ab[w].y];
          if (in) d = fmaf(q[w], sz, d);
          n += __popcll(__ballot(in && sz < 0.f));
        }
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) d += __shfl_xor(d, m);
      if (lane == 0) { a.cnt_out[chain0 + c] = n; a.diag_out[chain0 + c] = d; }
    }
  }
}

template <int NW, int R0, int R1>
hipError_t launch_t(hipStream_t s, const SweepArgs& a, const Sweep8Plan& sp) {
  const dim3 grid((a.B + 7) / 8), block(NW * 64);
#define SW8_LAUNCH(WL, ST)                                                                                 \
  do {                                                                                                     \
    hipError_t e = hipFuncSetAttribute((const void*)k_sweep8<NW, R0, R1, WL, ST>,                          \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp.lds);          \
    if (e != hipSuccess) return e;                                                                         \
    hipLaunchKernelGGL((k_sweep8<NW, R0, R1, WL, ST>), grid, block, sp.lds, s, a);                         \
    return hipGetLastError();                                                                              \
  } while (0)
  if (a.dbg_cycles) {
    if (sp.w1l) SW8_LAUNCH(true, true);
    SW8_LAUNCH(false, true);
  }
  if (sp.w1l) SW8_LAUNCH(true, false);
  SW8_LAUNCH(false, false);
#undef SW8_LAUNCH
}

}  // namespace

#ifndef SWEEP8_R1_256
#define SWEEP8_R1_256 0     // fragments of layer 1 kept in registers at 256 units
#endif

// the shapes k_sweep8 takes: plan_sweep8 (plan.hpp); the caller has checked it
hipError_t launch_sweep8(hipStream_t s, const SweepArgs& a, int Hp) {
  if (a.B <= 0) return hipSuccess;
  if (a.rbm || a.act != VMC_ACT_RELU_ || a.inj_up || a.dbg_up || a.acc_mask) return hipErrorInvalidValue;
  const Sweep8Plan sp = plan_sweep8(a.N, Hp, a.n_hidden, a.no_w1l != 0);
  if (!sp.ok) return hipErrorInvalidValue;
  if (Hp == 256) return launch_t<8, 32, SWEEP8_R1_256>(s, a, sp);
  return launch_t<4, 16, 16>(s, a, sp);
}
