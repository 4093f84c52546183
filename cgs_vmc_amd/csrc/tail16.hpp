// Row kernels of the dense ansaetze, templated on the hidden activation ACT
// (layers.NONLINEARITIES, layers.py:13-21):
//   k_tail16      layers 2..L + output for a list of rows (transposed, register resident)
//   k_tail0       the same without an H x H layer
//   k_backprop16  d logit / d z_l of every layer
// Included by act_tail.hip, which is compiled once per activation (-DVMC_INST_ACT=<id>).
#pragma once
#include "common.hpp"
#include <cstdlib>
#include <type_traits>

#ifndef TAIL_RD
#define TAIL_RD 8     // stages of k_tail16's weight-fragment ring (items of 8 MFMAs each)
#endif

// ---------------------------------------------------------------------------------- tail16
// Layers 2..L + output for a list of rows.  Persistent: one 256-thread workgroup per CU walks
// the row tiles b, b + gridDim.x, ...; each of its 4 waves owns 32 rows of a tile, as two 16-row
// halves that share every weight fragment (1 KiB of p16 per 8 v_mfma_f32_16x16x4_f32 = 256 matrix
// cycles) and give the matrix pipe two independent accumulator chains, and all Hp hidden units
// (2 x NT x 4 accumulator registers).  No barriers; LDS is only a per-wave staging area for the
// next tile's gathered first-layer activations.  RATIO mode writes
// 0.5*jx[bond]*exp(logit_row - logit_base[chain]).
// RBM: the last H x H layer's epilogue is sum_h log cosh(z_h) instead of relu(z) . w_out, and the
// onsite term x . w_on of the row (chain's cached value + the rank-2 exchange update) is added.
// Layouts (lane = 16 g + j): B operand of k-step e of input tile ti = X[row j][16 ti + 4 g + e],
// accumulator register r of output tile to = unit 16 to + 4 g + r of row j, so the accumulator
// of tile `to` IS the B operand of input tile `to` of the next layer (as in k_sweep16).
// (A v_mfma_f32_32x32x2_f32 version of this kernel measured 1 % slower; the sustained rate of
// this loop skeleton, tools/ubench/mfma_stream.hip, is 143 TFLOP/s = 91 % of nominal either way.)
template <int NT, bool RATIO, bool RBM, int ACT>
__global__ __launch_bounds__(256) void k_tail16(TailArgs a) {
  constexpr int Hp = NT * 16;
  static_assert(NT % 2 == 0, "the gather of an output tile is cut into two halves");
  // per-wave staging of the NEXT tile's first-layer activations: [wave][NT][2 halves][64 lanes]
  extern __shared__ float s_stage[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, j = lane & 15;
  f32x4* stage = (f32x4*)s_stage + wave * (2 * NT) * 64 + lane;
  const int n_rows = a.n_rows_dev ? *a.n_rows_dev : a.n_rows;
  const int n_hidden = a.n_hidden;   // >= 1 (no H x H layer goes through k_tail0)
  const float bout = a.pp.bout[0];
  const int oact = a.oact;

  // weight-fragment ring: item (to, ti) of a layer lives in stage item % RD and is issued RD-1
  // items ahead of its use, across layer AND tile boundaries; every issue is unconditional
  constexpr int NI = NT * NT, RD = TAIL_RD;
  static_assert(NI % RD == 0, "ring slots continue across layers only if RD divides the items per layer");
  f32x4 ring[RD];
  const unsigned lane_off = (unsigned)lane * 16u;
  const char* p16c = (const char*)a.pp.p16;   // re-made opaque every tile (an opaque per-tile copy keeps LICM from hoisting ~256 per-item 64-bit addresses, which spill)
  typedef const __attribute__((address_space(1))) char* gchar_p;
  typedef const __attribute__((address_space(1))) f32x4* gf32x4_p;
  auto issue = [&](int l, int item) {
    gchar_p base = (gchar_p)p16c + ((size_t)l * Hp * Hp + (size_t)item * 256) * sizeof(float);
    return *(gf32x4_p)(base + lane_off);
  };
#pragma unroll
  for (int i = 0; i < RD - 1; ++i) ring[i] = issue(0, i);

  // descriptor of one of this lane's two rows (half 0: row j, half 1: row 16 + j of the wave)
  struct Desc { const float* zb; const float* wa; const float* wb; float coef, on, lbase, hjx; int row, chain, bond, valid; };
  auto describe = [&](int tile, int half) {   // first two tiles of a wave only
    Desc d;
    d.row = tile * 128 + wave * 32 + 16 * half + j;
    d.valid = d.row < n_rows;
    const int2 ri = a.rowinfo[d.valid ? d.row : n_rows - 1];   // {chain, +-(bond+1) or 0}
    d.chain = ri.x;
    const int bs = ri.y;
    d.bond = (bs > 0 ? bs : -bs) - (bs != 0 ? 1 : 0);
    d.coef = bs > 0 ? -2.f : (bs < 0 ? 2.f : 0.f);          // -2 * s_i, 0 for a plain row
    const int2 ab = a.bonds[d.bond];
    d.wa = a.pp.w1p + (long long)ab.x * Hp;
    d.wb = a.pp.w1p + (long long)ab.y * Hp;
    d.zb = a.z1 + (long long)d.chain * Hp;
    d.on = 0.f;
    if (RBM) d.on = fmaf(d.coef, a.pp.won[ab.x] - a.pp.won[ab.y], a.on_base[d.chain]);
    d.lbase = RATIO ? a.logit_base[d.chain] : 0.f;
    d.hjx = RATIO ? a.half_jx[d.bond] : 0.f;
    return d;
  };
  auto finish_row = [&](const Desc& d, float part) {
    part += __shfl_xor(part, 16);    // (g0 + g1), (g2 + g3): a + b == b + a bit for bit
    part += __shfl_xor(part, 32);
    float logit = part + bout;
    if (RBM) logit += d.on;
    if (d.valid && g == 0) {
      if (RATIO) a.out[d.row] = d.hjx * vmc_out_ratio(oact, logit, d.lbase);
      else a.out[d.row] = logit;
    }
  };

  int tile = blockIdx.x;
  if (tile * 128 + wave * 32 >= n_rows) return;   // wave-uniform; no barriers in this kernel
  Desc cur[2], nxt_d[2];
  const int tile1 = (tile + (int)gridDim.x) * 128 + wave * 32 < n_rows ? tile + (int)gridDim.x : tile;
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) { cur[hf] = describe(tile, hf); nxt_d[hf] = describe(tile1, hf); }
  f32x4 in[2][NT];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int off = 16 * t + 4 * g;
      const f32x4 z = *(const f32x4*)(cur[hf].zb + off);
      const f32x4 x = *(const f32x4*)(cur[hf].wa + off);
      const f32x4 y = *(const f32x4*)(cur[hf].wb + off);
#pragma unroll
      for (int e = 0; e < 4; ++e) in[hf][t][e] = vmc_act<ACT>(fmaf(cur[hf].coef, x[e] - y[e], z[e]));
    }

  for (;;) {
    const int next_tile = tile + gridDim.x;
    const bool has_next = next_tile * 128 + wave * 32 < n_rows;   // wave-uniform
    int opaque0 = 0;
    asm volatile("" : "+s"(opaque0));   // keeps tile-invariant bias / w_out loads inside the loop
    asm volatile("" : "+s"(p16c));      // same for the per-item weight addresses
    // descriptors of the tile after next: built in three steps inside the last layer
    Desc nn_d[2];
    int2 nn_ri[2], nn_ab[2];
    float nn_onb[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) { nn_d[hf] = nxt_d[hf]; nn_ri[hf] = make_int2(0, 0); nn_ab[hf] = make_int2(0, 0); nn_onb[hf] = 0.f; }

    // ---- all but the last H x H layer: in -> out -> in
    for (int l = 0; l + 1 < n_hidden; ++l) {
      const float* __restrict__ bl = a.pp.bh + l * Hp + opaque0;
      f32x4 out[2][NT];
      f32x4 bias = *(const f32x4*)(bl + 4 * g);
#pragma unroll
      for (int to = 0; to < NT; ++to) {
        f32x4 acc0 = bias, acc1 = bias;
        if (to + 1 < NT) bias = *(const f32x4*)(bl + 16 * (to + 1) + 4 * g);
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
          const int item = to * NT + ti, nxt = item + RD - 1;
          if (nxt < NI) ring[nxt % RD] = issue(l, nxt);
          else ring[nxt % RD] = issue(l + 1, nxt - NI);
          __builtin_amdgcn_sched_barrier(0);   // keep the prefetch RD-1 items ahead
          const f32x4 w = ring[item % RD];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], in[0][ti][e], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], in[1][ti][e], acc1, 0, 0, 0);
          }
        }
        out[0][to] = acc0; out[1][to] = acc1;
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) in[hf][t][e] = vmc_act<ACT>(out[hf][t][e]);
    }

    // ---- last H x H layer fused with the output dot.  While output tile `to` is accumulated,
    // unit tile `to` of the NEXT row tile is gathered: half 0 in the first NT/2 items (loads at
    // the first, arithmetic + LDS store at the last), half 1 in the second NT/2.
    float part[2] = {0.f, 0.f};
    {
      const int l = n_hidden - 1;
      const float* __restrict__ bl = a.pp.bh + l * Hp + opaque0;
      const float* __restrict__ wop = a.pp.woutp + opaque0;
      constexpr int SEG = NT / 2;
      f32x4 bias = *(const f32x4*)(bl + 4 * g);
#pragma unroll
      for (int to = 0; to < NT; ++to) {
        f32x4 wo;
        if (!RBM) wo = *(const f32x4*)(wop + 16 * to + 4 * g);
        f32x4 acc0 = bias, acc1 = bias;
        if (to + 1 < NT) bias = *(const f32x4*)(bl + 16 * (to + 1) + 4 * g);
        f32x4 gz, gx, gy;
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
          const int hf = ti / SEG;
          if (ti % SEG == 0) {
            const int off = 16 * to + 4 * g;
            gz = *(const f32x4*)(nxt_d[hf].zb + off);
            gx = *(const f32x4*)(nxt_d[hf].wa + off);
            gy = *(const f32x4*)(nxt_d[hf].wb + off);
          }
          const int item = to * NT + ti, nxt = item + RD - 1;
          if (item == 0) {                      // step A: rowinfo of the tile after next
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              nn_d[q].row = (next_tile + (int)gridDim.x) * 128 + wave * 32 + 16 * q + j;
              nn_d[q].valid = nn_d[q].row < n_rows;
              nn_ri[q] = a.rowinfo[nn_d[q].valid ? nn_d[q].row : n_rows - 1];
            }
          }
          if (item == NI / 4) {                 // step B: chain / bond -> bond table, z1 row
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              nn_d[q].chain = nn_ri[q].x;
              const int bs = nn_ri[q].y;
              nn_d[q].bond = (bs > 0 ? bs : -bs) - (bs != 0 ? 1 : 0);
              nn_d[q].coef = bs > 0 ? -2.f : (bs < 0 ? 2.f : 0.f);
              nn_ab[q] = a.bonds[nn_d[q].bond];
              nn_d[q].zb = a.z1 + (long long)nn_d[q].chain * Hp;
              if (RBM) nn_onb[q] = a.on_base[nn_d[q].chain];
              nn_d[q].lbase = RATIO ? a.logit_base[nn_d[q].chain] : 0.f;
              nn_d[q].hjx = RATIO ? a.half_jx[nn_d[q].bond] : 0.f;
            }
          }
          if (item == NI / 2) {                 // step C: W1 rows of the exchanged sites
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              nn_d[q].wa = a.pp.w1p + (long long)nn_ab[q].x * Hp;
              nn_d[q].wb = a.pp.w1p + (long long)nn_ab[q].y * Hp;
              nn_d[q].on = 0.f;
              if (RBM) nn_d[q].on = fmaf(nn_d[q].coef, a.pp.won[nn_ab[q].x] - a.pp.won[nn_ab[q].y], nn_onb[q]);
            }
          }
          if (nxt < NI) ring[nxt % RD] = issue(l, nxt);
          else ring[nxt % RD] = issue(0, nxt - NI);          // next tile's first layer
          __builtin_amdgcn_sched_barrier(0);
          const f32x4 w = ring[item % RD];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], in[0][ti][e], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], in[1][ti][e], acc1, 0, 0, 0);
          }
          if (ti % SEG == SEG - 1) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = vmc_act<ACT>(fmaf(nxt_d[hf].coef, gx[e] - gy[e], gz[e]));
            stage[(to * 2 + hf) * 64] = v;
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (RBM) {
            const bool ok = 16 * to + 4 * g + e < a.n_units;
            part[0] += ok ? vmc_logcosh(acc0[e]) : 0.f;
            part[1] += ok ? vmc_logcosh(acc1[e]) : 0.f;
          } else {
            part[0] = fmaf(vmc_act<ACT>(acc0[e]), wo[e], part[0]);
            part[1] = fmaf(vmc_act<ACT>(acc1[e]), wo[e], part[1]);
          }
        }
      }
    }
    finish_row(cur[0], part[0]);
    finish_row(cur[1], part[1]);
    if (!has_next) break;
    tile = next_tile;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) { cur[hf] = nxt_d[hf]; nxt_d[hf] = nn_d[hf]; }
    // the wave re-reads what it wrote itself (same lane, same address): no barrier needed
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int t = 0; t < NT; ++t) in[hf][t] = stage[(t * 2 + hf) * 64];
  }
}

// rowinfo of a plain batch of rows: row r = chain r, no exchange update


// no H x H layer: logit = f(z1') . w_out + b_out (FC, L = 1) or sum log cosh(z1') + onsite
// (RBM, num_layers = 0: the classic restricted Boltzmann machine).  One WAVE per row: the lanes read
// the chain's cached z1 and the two W1 rows of the exchanged sites as consecutive 16-byte vectors
// (1 KiB per load instruction), every lane sums its Hp / 64 terms and the wave folds the 64 partials
// in a fixed tree (in double: the RBM logit is a sum of H positive terms of order 1 whose
// DIFFERENCES between configurations are what the sampler and the local energy use).  L2-bound:
// 3 Hp floats per row; any Hp that is a multiple of 64.
template <bool RATIO, bool RBM, int ACT>
__global__ __launch_bounds__(256) void k_tail0(TailArgs a, int Hp) {
  const int n_rows = a.n_rows_dev ? *a.n_rows_dev : a.n_rows;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int n_waves = gridDim.x * 4;
  for (int row = wave; row < n_rows; row += n_waves) {
    int chain = row, bs = 0;
    if (a.rowinfo) { const int2 ri = a.rowinfo[row]; chain = ri.x; bs = ri.y; }
    const float* zb = a.z1 + (long long)chain * Hp;
    const float* wa = a.pp.w1p;
    const float* wb = wa;
    float coef = 0.f, on = RBM ? a.on_base[chain] : 0.f;
    int bond = 0;
    if (bs != 0) {
      bond = (bs > 0 ? bs : -bs) - 1;
      coef = bs > 0 ? -2.f : 2.f;
      const int2 ab = a.bonds[bond];
      wa += (long long)ab.x * Hp; wb += (long long)ab.y * Hp;
      if (RBM) on = fmaf(coef, a.pp.won[ab.x] - a.pp.won[ab.y], on);
    }
    float part = 0.f;
    for (int i = 4 * lane; i < Hp; i += 256) {
      const f32x4 z = *(const f32x4*)(zb + i);
      const f32x4 x = *(const f32x4*)(wa + i);
      const f32x4 y = *(const f32x4*)(wb + i);
      f32x4 w = f32x4{0.f, 0.f, 0.f, 0.f};
      if (!RBM) w = *(const f32x4*)(a.pp.woutp + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = fmaf(coef, x[e] - y[e], z[e]);
        if (RBM) part += i + e < a.n_units ? vmc_logcosh(v) : 0.f;
        else part = fmaf(vmc_act<ACT>(v), w[e], part);
      }
    }
    double s = (double)part;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if (lane == 0) {
      const float logit = (float)s + a.pp.bout[0] + on;
      a.out[row] = RATIO ? a.half_jx[bond] * vmc_out_ratio(a.oact, logit, a.logit_base[chain]) : logit;
    }
  }
}

template <int NT, bool RATIO, bool RBM, int ACT>
static hipError_t launch_tail16_h(hipStream_t s, const TailArgs& a) {
  const int tiles = (a.n_rows + 127) / 128;
  const int persistent = a.num_cus > 0 ? a.num_cus : 256;   // 1 workgroup per CU (1 wave/SIMD)
  const dim3 grid(tiles < persistent ? tiles : persistent), block(256);
  const size_t lds = (size_t)4 * (2 * NT) * 64 * sizeof(f32x4);
  hipError_t e = hipFuncSetAttribute((const void*)k_tail16<NT, RATIO, RBM, ACT>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_tail16<NT, RATIO, RBM, ACT>), grid, block, lds, s, a);
  return hipGetLastError();
}

template <bool RATIO, bool RBM, int ACT>
static hipError_t launch_tail_t(hipStream_t s, const TailArgs& a, int Hp) {
  if (a.n_rows <= 0) return hipSuccess;
  if (a.n_hidden == 0) {
    const int blocks = (a.n_rows + 3) / 4;      // one wave per row, 8 workgroups per CU's worth of rows in flight
    hipLaunchKernelGGL((k_tail0<RATIO, RBM, ACT>), dim3(blocks < 4096 ? blocks : 4096), dim3(256), 0, s, a, Hp);
    return hipGetLastError();
  }
  switch (Hp / 16) {
    case 4: return launch_tail16_h<4, RATIO, RBM, ACT>(s, a);
    case 8: return launch_tail16_h<8, RATIO, RBM, ACT>(s, a);
    case 12: return launch_tail16_h<12, RATIO, RBM, ACT>(s, a);
    case 16: return launch_tail16_h<16, RATIO, RBM, ACT>(s, a);
    default: return hipErrorInvalidValue;
  }
}


// ------------------------------------------------------------------------------- backprop16
// d logit / d z_l for every layer (training.py:545-547 asks tf.gradients for it), 16 chains per
// workgroup like the sampler: delta_last = w_out (.) relu'(a_last) (FC) or tanh(z_last) (RBM,
// already in the last activation slot), then delta_{l-1} = (delta_l W_l^T) (.) relu'(a_{l-1})
// on 16x16x4 MFMA with the transposed weight image p16t; the operand of the next layer goes
// through LDS (one barrier per layer), the result to global memory row-major for the
// weight-gradient GEMMs.  One pass per accumulate call: 2 x 128 MFMAs per wave at H = 256.
// General activation f: relu' becomes f'(z), read off the stored activation a = f(z) (or, for the
// cosine, from the dact array its producer wrote); `oscale` (nullable) is the per-sample factor
// (1/psi) d psi / d x of a non-exp output activation.
template <int NT, int NW, int ACT>
__global__ __launch_bounds__(NW * 64) void k_backprop16(const float* __restrict__ act_all,
                                                        float* __restrict__ delta_all,
                                                        const float* __restrict__ p16t,
                                                        const float* __restrict__ woutp, int B,
                                                        int n_hidden, int rbm,
                                                        const float* __restrict__ dact_all,
                                                        const float* __restrict__ oscale, ElocFold ef,
                                                        OutLayerSums op) {
#ifndef VMC_BACKPROP_PF
#define VMC_BACKPROP_PF 8
#endif
  constexpr int Hp = NT * 16, TO = NT / NW, PF = (NT % VMC_BACKPROP_PF == 0) ? VMC_BACKPROP_PF : 4;   // weight-ring depth
  static_assert(NT % NW == 0 && NT % PF == 0, "tiles divide over waves and the prefetch ring");
  __shared__ __attribute__((aligned(16))) float s_x[2 * NT * 256];   // [2][NT][64 lanes][4]
  __shared__ float s_wj[16];                                         // per-chain weight of the second sum (op.part)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, j = lane & 15;
  const int chain = blockIdx.x * 16 + j;
  const bool ok = chain < B;
  const long long row = (long long)(ok ? chain : 0) * Hp;
  const long long layer_stride = (long long)B * Hp;
  // f'(z) of layer l at this lane's 4 units of tile t
  auto fprime = [&](int l, int col, const f32x4& a) {
    f32x4 d;
    if (ACT == VMC_ACT_COS_) d = *(const f32x4*)(dact_all + l * layer_stride + row + col);
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = vmc_dact_from_a<ACT>(a[e]);
    }
    return d;
  };
  const float osc = (oscale && ok) ? oscale[chain] : 1.f;
  // local energies of this workgroup's 16 chains (k_eloc_reduce's job and order: lane-strided partial
  // sums, xor tree): wave w takes chains 16 / NW at a time.  Nothing below reads them.
  // The wave's chains are taken TOGETHER and without a branch around a load: both segment bounds first,
  // then the first 128 rows of every chain (clamped addresses, masked values), so that the 16 / NW chains
  // cost two dependent round trips, not two each; a chain's rows beyond 128 follow in the general loop.
  if (ef.off) {
    constexpr int Q = 16 / NW;
    int cq[Q], r0[Q], r1[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int c = blockIdx.x * 16 + wave * Q + q;
      cq[q] = c < B ? c : B - 1;
      r0[q] = ef.off[cq[q]];
      r1[q] = ef.off[cq[q] + 1];
    }
    float v0[Q], v1[Q], dg[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int last = r1[q] > r0[q] ? r1[q] - 1 : r0[q];       // (an empty segment reads one row of its neighbour: masked)
      const int ra = r0[q] + lane, rb = ra + 64;
      v0[q] = ef.val[ra < last ? ra : last];
      v1[q] = ef.val[rb < last ? rb : last];
      dg[q] = ef.diag[cq[q]];
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int c = blockIdx.x * 16 + wave * Q + q;
      const int ra = r0[q] + lane, rb = ra + 64;
      float sum = 0.f;
      if (ra < r1[q]) sum += v0[q];
      if (rb < r1[q]) sum += v1[q];
      for (int r = rb + 64; r < r1[q]; r += 64) sum += ef.val[r];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
      if (lane == 0 && c < B) {
        if (ef.offdiag) ef.offdiag[c] = sum;
        const float e = dg[q] + sum;
        ef.eloc[c] = e;
        float wgt = e;
        if (ef.ratio) {     // LogOverlapITSWO: the supervisor's local energy -> the ratio (k_itswo_ratio's expression)
          const float lw = ef.logit_omega[c], lp = ef.logit_psi[c];
          const float amp = ef.oact == VMC_ACT_EXP_ ? expf(lw - lp + ef.log_factor)
                                                    : vmc_act_rt(ef.oact, lw) / vmc_act_rt(ef.oact, lp);
          wgt = amp * (1.f - ef.beta * e);
          ef.ratio[c] = wgt;
        }
        if (op.part && op.w == (ef.ratio ? ef.ratio : ef.eloc)) s_wj[wave * Q + q] = wgt;
      }
    }
  }
  // last layer's delta for this wave's own unit tiles
  {
    const float* a_last = act_all + n_hidden * layer_stride + row;
    float* d_last = delta_all + n_hidden * layer_stride + row;
#pragma unroll
    for (int to = 0; to < TO; ++to) {
      const int t = wave * TO + to, col = 16 * t + 4 * g;
      const f32x4 a = *(const f32x4*)(a_last + col);
      f32x4 d;
      if (rbm) d = a;
      else {
        const f32x4 w = *(const f32x4*)(woutp + col);
        const f32x4 fp = fprime(n_hidden, col, a);
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = osc * (w[e] * fp[e]);
      }
      if (!ok) d = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ok) *(f32x4*)(d_last + col) = d;
      *(f32x4*)(s_x + (t * 64 + lane) * 4) = d;
    }
  }
  int cur = 0;
  for (int l = n_hidden; l > 0; --l) {
    __syncthreads();
    const f32x4* __restrict__ wp = (const f32x4*)(p16t + (long long)(l - 1) * Hp * Hp) + lane;
    const f32x4* xin = (const f32x4*)(s_x + cur * NT * 256) + lane;
    const float* a_prev = act_all + (l - 1) * layer_stride + row;
    f32x4 mask[TO];
#pragma unroll
    for (int to = 0; to < TO; ++to) mask[to] = *(const f32x4*)(a_prev + 16 * (wave * TO + to) + 4 * g);
    f32x4 acc[TO], wb[PF][TO];
#pragma unroll
    for (int to = 0; to < TO; ++to) acc[to] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < PF - 1; ++st)
#pragma unroll
      for (int to = 0; to < TO; ++to) wb[st][to] = wp[((wave * TO + to) * NT + st) * 64];
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
      const int tn = ti + PF - 1 < NT ? ti + PF - 1 : NT - 1;   // clamped: unconditional issue
#pragma unroll
      for (int to = 0; to < TO; ++to) wb[(ti + PF - 1) % PF][to] = wp[((wave * TO + to) * NT + tn) * 64];
      const f32x4 b = xin[ti * 64];
      // (the loads stay PF - 1 tiles ahead of their use: unpinned, the compiler sinks each next to its MFMAs and
      // every k-tile waits an L2 round trip -- the kernel ran at 0.25 MFMA busy, half its wave cycles waiting)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int to = 0; to < TO; ++to)
          acc[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[ti % PF][to][r], b[r], acc[to], 0, 0, 0);
    }
    float* d_prev = delta_all + (l - 1) * layer_stride + row;
    float* xout = s_x + (cur ^ 1) * NT * 256;
#pragma unroll
    for (int to = 0; to < TO; ++to) {
      const int t = wave * TO + to;
      f32x4 d;
      const f32x4 fp = fprime(l - 1, 16 * t + 4 * g, mask[to]);
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = ok ? acc[to][e] * fp[e] : 0.f;
      if (ok) *(f32x4*)(d_prev + 16 * t + 4 * g) = d;
      *(f32x4*)(xout + (t * 64 + lane) * 4) = d;
    }
    cur ^= 1;
  }
  // Output layer's weight gradient over this workgroup's chains (OutLayerSums, common.hpp), LAST: the weight
  // of the second sum is the local energy folded at the top (through LDS -- the layer loop's barriers, or the
  // one below, lie in between, so nothing waits for the fold's round trips) or a vector of an earlier launch;
  // the last activations are read again (L2).
  if (op.part) {                                   // block-uniform
    float wj;
    if (ef.off && op.w == (ef.ratio ? ef.ratio : ef.eloc)) {
      if (n_hidden == 0) __syncthreads();
      wj = ok ? s_wj[j] : 0.f;
    } else {
      wj = ok ? op.w[chain] : 0.f;
    }
    // sum over the 16 lanes of a DPP row (the 16 chains): xor 1, xor 2, half mirror, mirror -- every lane
    // ends with the same value
    auto row16_sum = [](float v) {
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));
      return v;
    };
    constexpr int LDP = Hp + 4;
    float* part = op.part + (long long)blockIdx.x * 2 * LDP;
    if (wave == 0) {                               // the bias: sum_b s_b and sum_b w_b s_b
      const float s1 = row16_sum(ok ? osc : 0.f), s2 = row16_sum(ok ? osc * wj : 0.f);
      if (lane == 0) { part[Hp] = s1; part[LDP + Hp] = s2; }
    }
    const float* a_last = act_all + n_hidden * layer_stride + row;
#pragma unroll
    for (int to = 0; to < TO; ++to) {
      const int col = 16 * (wave * TO + to) + 4 * g;
      const f32x4 a = *(const f32x4*)(a_last + col);
      f32x4 s1, s2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = ok ? osc * a[e] : 0.f;
        s1[e] = row16_sum(v); s2[e] = row16_sum(v * wj);
      }
      if (j == 0) { *(f32x4*)(part + col) = s1; *(f32x4*)(part + LDP + col) = s2; }
    }
  }
}



template <int ACT>
static hipError_t launch_backprop16_t(hipStream_t s, const float* act_all, float* delta_all,
                                      const float* p16t, const float* woutp, int B, int Hp,
                                      int n_hidden, bool rbm, const float* dact_all,
                                      const float* oscale, const ElocFold& ef, const OutLayerSums& op) {
  if (B <= 0) return hipSuccess;
  const dim3 grid((B + 15) / 16);
  const int r = rbm ? 1 : 0;
  switch (Hp / 16) {
    case 4: hipLaunchKernelGGL((k_backprop16<4, 4, ACT>), grid, dim3(256), 0, s, act_all, delta_all, p16t, woutp, B, n_hidden, r, dact_all, oscale, ef, op); break;
    case 8: hipLaunchKernelGGL((k_backprop16<8, 4, ACT>), grid, dim3(256), 0, s, act_all, delta_all, p16t, woutp, B, n_hidden, r, dact_all, oscale, ef, op); break;
    case 12: hipLaunchKernelGGL((k_backprop16<12, 4, ACT>), grid, dim3(256), 0, s, act_all, delta_all, p16t, woutp, B, n_hidden, r, dact_all, oscale, ef, op); break;
    case 16: hipLaunchKernelGGL((k_backprop16<16, 8, ACT>), grid, dim3(512), 0, s, act_all, delta_all, p16t, woutp, B, n_hidden, r, dact_all, oscale, ef, op); break;
    // 257 .. 512 hidden units (384 / 512 padded): wave w owns 3 or 4 output tiles, 48 / 64 KB of operands in LDS
    case 24: hipLaunchKernelGGL((k_backprop16<24, 8, ACT>), grid, dim3(512), 0, s, act_all, delta_all, p16t, woutp, B, n_hidden, r, dact_all, oscale, ef, op); break;
    case 32: hipLaunchKernelGGL((k_backprop16<32, 8, ACT>), grid, dim3(512), 0, s, act_all, delta_all, p16t, woutp, B, n_hidden, r, dact_all, oscale, ef, op); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <int ACT>
static hipError_t launch_tail_act(hipStream_t s, const TailArgs& a, int Hp, bool ratio_mode, bool rbm) {
  if (rbm) return ratio_mode ? launch_tail_t<true, true, ACT>(s, a, Hp) : launch_tail_t<false, true, ACT>(s, a, Hp);
  return ratio_mode ? launch_tail_t<true, false, ACT>(s, a, Hp) : launch_tail_t<false, false, ACT>(s, a, Hp);
}
