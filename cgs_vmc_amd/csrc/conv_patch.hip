// k_cgen_patch_sweep: the sampler of the GENERAL convolution path for lattices much larger than a convolution's reach
// (round 6).  The reference evaluates the whole network on every proposed configuration (graph_builders.py:57-88 maps
// `wavefunction(configs)` over the batch, wavefunctions.py:534-579), and so did every sampler here: a full forward of the
// B candidates per mc_step -- at 36 x 36 sites, 3 x 16 filters 5 x 5 and 32 chains four launches of latency, 52 us per step.
// But an exchange (graph_builders.py:67-71) negates TWO spins, and a periodic convolution with K taps per axis
// (layers.py:118-160: out(p) = sum_d w[d] in(p + d - lo)) carries a changed input site q only to the outputs
// [q - hi, q + lo]: convolution l of the network changes inside a box of (l + 1)(K - 1) + 1 sites per axis around each of
// the two sites and nowhere else.  So:
//   * the maps of every convolution of every chain stay in HBM ([n_conv][B][N][Fp], filled by one taped full forward per
//     launch);
//   * ONE workgroup per chain runs n_steps steps in one launch.  Per step and convolution it stages the input window of each
//     of the two boxes in LDS -- the chain's stored map of the convolution below, overlaid with that convolution's two
//     freshly computed boxes -- and computes the box with k_cgen_band's tile arithmetic (16 channels x 16 positions per
//     v_mfma_f32_16x16x4_f32, taps outer, channels inner, the same order: a box value has the bits the full forward gives);
//   * the candidate's logit is the sum of the last map with its two boxes overlaid, in k_cgen_step_tail's order (thread-
//     strided doubles, xor tree, the four waves in order): the Metropolis test sees the bits of the full-forward sampler,
//     so the chains are the same chains, bit for bit (tests/test_gpu_conv_general.py);
//   * an accepted move writes the boxes into the stored maps; the next proposal is k_cgen_step_tail's.
// Work per step: 2 sum_l box_l^2 positions instead of n_conv N -- 36 x 36, 3 convolutions 5 x 5: 550 of 3,888 -- and no
// launch, no grid-wide hand-over between the convolutions: a chain's step is a workgroup's business.
// Residual networks (layers.py:226-228, wavefunctions.py:766-772): the initial convolution is linear, a block's first ends in
// selu, its second adds the block's input h -- read from the map two convolutions below with ITS boxes overlaid.
// Shapes: plan_cgen_patch_ok (plan.hpp).
#include "conv.hpp"

#include <cstdlib>
#include <type_traits>

namespace {

__device__ __forceinline__ float cp_pre(int pre, float x) { return pre < 0 ? x : vmc_act_rt(pre, x); }
__device__ __forceinline__ float cp_selu(float x) {   // layers.py:226 tf.nn.selu (k_cgen_band's form and constants)
  const float scale = 1.0507009873554805f, alpha = 1.6732632423543772f;
  return scale * (x > 0.f ? x : alpha * (expf(x) - 1.f));
}
// v in [-2 d, d): a site coordinate minus a box / window offset (at most the last box's side + K - 1 <= 2 d) -- no division
__device__ __forceinline__ int cp_wrap(int v, int d) { v += v < 0 ? d : 0; v += v < 0 ? d : 0; return v; }
// v in [0, 3 d): the coordinate of a window site = a wrapped origin + an offset below d + K
__device__ __forceinline__ int cp_fold(int v, int d) { v -= v >= d ? d : 0; v -= v >= d ? d : 0; return v; }
// a - b for a, b in [0, d), wrapped
__device__ __forceinline__ int cp_rel(int a, int b, int d) { const int r = a - b; return r < 0 ? r + d : r; }
// x / d for 0 <= x < 2^20, 1 <= d <= 2^10 with inv = 1.f / d (exact: the quotient's error stays below half a step of 1 / d)
__device__ __forceinline__ int cp_div(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

// NW: waves per workgroup -- 4 (one per SIMD: up to 512 registers, what 6 x 6 and 7 x 7 taps of fragments need) or 8 (two
// per SIMD: the phases of a step are chains of dependent work -- a second wave fills their gaps).  The map sum is taken
// by the first four waves either way: its order is k_cgen_step_tail's, 256 threads.
// ELOC: the same boxes for the LOCAL ENERGIES (operators.py:162-169: a connected configuration is its chain with one
// antiparallel bond exchanged -- two spins again).  The workgroups walk the rows of a row list (chain, bond) instead of
// the steps of one chain: the chain's stored maps are read only, nothing is tested or committed, and the sum of the last
// map with its boxes overlaid -- k_cgen_rowsum's double, in its order -- goes to out_sum[row].
template <int K, int KW, int NW, bool ELOC>
__global__ __launch_bounds__(64 * NW, 1) void k_cgen_patch_sweep(CgenPatchArgs a) {
  constexpr int T = K * KW, NF0 = (T + 3) / 4, NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ double s_w[4];
  __shared__ int s_prop[3];                    // the site raised (iup), the site lowered (idn), accepted
  __shared__ float s_u;
  __shared__ float s_bv[2][8];                 // the proposal's best values / sites of the waves
  __shared__ int s_bi[2][8];
  const ConvGeom g = a.g;
  const int D1 = g.D1, D2 = g.D2, N = g.N, F = g.F, Fp = a.Fp, L = g.n_conv, FQ = Fp >> 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, gq = lane >> 4;
  long long c = blockIdx.x;                    // the chain (ELOC: of the row at hand)
  // ELOC: a bond's two sites are neighbours, their boxes overlap almost entirely -- ONE box per convolution then, the two
  // boxes' bounding box: e1 x e2 sites larger (the sites' displacement), anchored at the first site along each axis; its
  // positions outside both boxes recompute what is stored.  It lives in the space of the two (nbx = 1).
  int e1 = 0, e2 = 0, nbx = 2;
  auto base1 = [&](int l) { return (l + 1) * (K - 1) + 1; };
  auto base2 = [&](int l) { return (l + 1) * (KW - 1) + 1; };
  auto side1 = [&](int l) { return base1(l) + (ELOC ? e1 : 0); };
  auto side2 = [&](int l) { return base2(l) + (ELOC ? e2 : 0); };
  auto poff = [&](int l) { int o = 0; for (int j = 0; j < l; ++j) o += 2 * base1(j) * base2(j) * 16; return o; };
#define NBX (ELOC ? nbx : 2)
  // ---- LDS (plan_cgen_patch_lds_bytes): spins, weight fragments, biases, the boxes of every convolution, two windows
  float* const s_x = sm;
  float* const s_wf = s_x + ((N + 3) & ~3);
  float* const s_bias = s_wf + (L - 1) * T * 256;
  float* const s_patch = s_bias + L * 16;
  float* const s_win = s_patch + poff(L);
  int wstride = (2 * K - 1) * (2 * KW - 1);
  for (int l = 1; l < L; ++l) wstride = max(wstride, (base1(l) + K - 1) * (base2(l) + KW - 1) * 16);
  short* const s_ovl = (short*)(s_win + 2 * wstride);     // [N]: the place of a site in the last convolution's boxes, or -1
  float* const s_ru = s_win + 2 * wstride + (N + 7) / 8 * 4;  // [N]: the next step's site uniforms (eight waves: drawn under the map sum)
  __shared__ float s_unext;                               // ... and its acceptance uniform

  // ---- once per launch: the chain, the parameters
  const long long w_base1 = (long long)T * F + F, w_per = (long long)T * F * F + F;     // theta: w_0, b_0, then (w_l, b_l)
  for (int i = tid; i < N; i += NT) { if (!ELOC) s_x[i] = a.configs[c * N + i]; s_ovl[i] = -1; }
  for (int i = tid; i < (L - 1) * T * 256; i += NT) {
    // fragment (l, tap t) of lane ln, MFMA e: the A operand of k_cgen_band -- output channel ln & 15 against input channel 4 (ln >> 4) + e
    const int l1 = i / (T * 256), r = i - l1 * T * 256, t = r >> 8, ln = (r >> 2) & 63, e = r & 3;
    const int c_in = 4 * (ln >> 4) + e, fo = ln & 15;
    const float* w = a.theta + w_base1 + l1 * w_per;
    s_wf[i] = (c_in < F && fo < F) ? w[((long long)t * F + c_in) * F + fo] : 0.f;
  }
  for (int i = tid; i < L * 16; i += NT) {
    const int l = i >> 4, f = i & 15;
    const float* b = l == 0 ? a.theta + (long long)T * F : a.theta + w_base1 + (l - 1) * w_per + (long long)T * F * F;
    s_bias[i] = f < F ? b[f] : 0.f;
  }
  float wf0[NF0];                              // first convolution: the taps run over the MFMA's k index (k_cgen_band<FIRST>)
  int tr0[NF0], tc0[NF0];                      // ... this lane's tap of MFMA m: its row and column in the window
#pragma unroll
  for (int m = 0; m < NF0; ++m) {
    const int t = 4 * m + gq, tt = t < T ? t : 0;
    wf0[m] = (t < T && p < F) ? a.theta[(long long)t * F + p] : 0.f;
    tr0[m] = tt / KW; tc0[m] = tt % KW;
  }
  float cur_logit = 0.f;
  unsigned n_acc = 0;
  if (!ELOC && tid == 0) { s_prop[0] = a.iup[c]; s_prop[1] = a.idn[c]; s_u = a.u[c]; cur_logit = a.logit[c]; }
  __syncthreads();

  const int pre = a.post ? -1 : a.act;
  const float inv_d2_ = 1.f / (float)D2;
  // phase clocks of chain 0 (diagnostic: CgenPatchArgs.prof; s_memtime ticks summed over the steps): 0 first convolution,
  // 1 staging of the others, 2 their tiles, 3 the map sum, 4 test + commit, 5 the next proposal
  unsigned long long t_prev = 0, t_ph[6] = {0, 0, 0, 0, 0, 0};
  const bool prof = a.prof != nullptr && blockIdx.x == 0 && tid == 0;
  auto stamp = [&](int ph) {
    if (prof) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_ph[ph] += t - t_prev; t_prev = t; }
  };
  if (prof) t_prev = __builtin_amdgcn_s_memtime();
  for (long long st = ELOC ? (long long)blockIdx.x : 0; st < (ELOC ? a.n_rows : a.n_steps); st += ELOC ? (long long)gridDim.x : 1) {
    int up, dn;
    bool flip = true;
    if (ELOC) {       // the row: its chain, and the bond whose two sites are exchanged (0: the chain itself -- its boxes are what is stored)
      const int2 ri = a.rowinfo[a.row0 + st];
      c = __builtin_amdgcn_readfirstlane(ri.x);
      const int bi = __builtin_amdgcn_readfirstlane(ri.y);
      flip = bi != 0;
      const int2 ab = a.bonds[(bi > 0 ? bi : -bi) - (flip ? 1 : 0)];
      up = flip ? __builtin_amdgcn_readfirstlane(ab.x) : 0;
      dn = flip ? __builtin_amdgcn_readfirstlane(ab.y) : (N > 1 ? 1 : 0);
    } else {
      up = __builtin_amdgcn_readfirstlane(s_prop[0]); dn = __builtin_amdgcn_readfirstlane(s_prop[1]);   // (uniform: scalar arithmetic below)
    }
    int q1[2] = {cp_div(up, inv_d2_), cp_div(dn, inv_d2_)};
    int q2[2] = {up - q1[0] * D2, dn - q1[1] * D2};
    if (ELOC) {
      int dy = q1[1] - q1[0], dx = q2[1] - q2[0];           // the second site from the first, the shorter way round
      dy += dy > D1 / 2 ? -D1 : (dy < -((D1 - 1) / 2) ? D1 : 0);
      dx += dx > D2 / 2 ? -D2 : (dx < -((D2 - 1) / 2) ? D2 : 0);
      const int ady = dy < 0 ? -dy : dy, adx = dx < 0 ? -dx : dx;
      const bool merged = ady <= 1 && adx <= 1 && base1(L - 1) + ady <= D1 && base2(L - 1) + adx <= D2 &&
                          (K + ady) * (KW + adx) <= 2 * K * KW;       // (the first convolution's box is the tightest fit)
      e1 = merged ? ady : 0; e2 = merged ? adx : 0; nbx = merged ? 1 : 2;
      if (merged) { q1[0] = dy >= 0 ? q1[0] : q1[1]; q2[0] = dx >= 0 ? q2[0] : q2[1]; }
    }
    // ---- convolution 0: the windows are the candidate's spins (the exchanged pair negated)
    {
      const int s1 = side1(0), s2 = side2(0), SC = s2 + KW - 1, n_win = (s1 + K - 1) * SC, n_pos = s1 * s2, n_tiles = (n_pos + 15) >> 4;
      const float inv_sc = 1.f / (float)SC, inv_s2 = 1.f / (float)s2;
      const int o1[2] = {cp_wrap(q1[0] - g.hi - g.lo, D1), cp_wrap(q1[1] - g.hi - g.lo, D1)};
      const int o2[2] = {cp_wrap(q2[0] - g.hi2 - g.lo2, D2), cp_wrap(q2[1] - g.hi2 - g.lo2, D2)};
      for (int i = tid; i < NBX * n_win; i += NT) {
        const int b = i >= n_win, j = i - b * n_win, wy = cp_div(j, inv_sc), wx = j - wy * SC;
        const int site = cp_fold((b ? o1[1] : o1[0]) + wy, D1) * D2 + cp_fold((b ? o2[1] : o2[0]) + wx, D2);
        const float x = ELOC ? a.configs[c * N + site] : s_x[site];
        s_win[b * wstride + j] = (flip && (site == up || site == dn)) ? -x : x;
      }
      __syncthreads();
      f32x4 bias4;
#pragma unroll
      for (int r = 0; r < 4; ++r) bias4[r] = s_bias[4 * gq + r];
      for (int tt = wave; tt < NBX * n_tiles; tt += NW) {
        const int b = tt >= n_tiles, tile = tt - b * n_tiles;
        const int q = tile * 16 + p, qq = q < n_pos ? q : n_pos - 1;
        const int y = cp_div(qq, inv_s2), x = qq - y * s2;
        f32x4 acc = bias4;
        const float* base = s_win + b * wstride + y * SC + x;
#pragma unroll
        for (int m = 0; m < NF0; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf0[m], base[tr0[m] * SC + tc0[m]], acc, 0, 0, 0);
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc));    // the relu of vmc_act_rt is an asm v_max_f32 (common.hpp: vmc_mfma_settle)
        f32x4 v = acc;
        if (a.post && !g.resnet) {          // (ResNet2D: the initial convolution is linear, wavefunctions.py:766)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = vmc_act_rt(a.act, v[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 4 * gq + e < F ? v[e] : 0.f;
        if (q < n_pos) *(f32x4*)(s_patch + b * n_pos * 16 + q * 16 + 4 * gq) = v;
      }
    }
    // the first SU quads per thread of the last map are requested before the last convolution's tiles and added behind them
    constexpr int SU = NW == 8 ? 8 : (T > 25 ? 12 : 24);
    constexpr bool SUM_AHEAD = NW == 4;      // (eight waves: 256 registers per wave)
    f32x4 vs[SU];
    const int nq = N * FQ;
    const float* const lmap = a.maps + (L - 1) * a.map_stride + c * N * Fp;
    // ---- the convolutions behind it
    for (int l = 1; l < L; ++l) {
      __syncthreads();                 // the boxes of convolution l - 1 are written; the windows' readers are done
      stamp(l == 1 ? 0 : 2);
      const int s1 = side1(l), s2 = side2(l), SC = s2 + KW - 1, n_win = (s1 + K - 1) * SC;
      const int ps1 = side1(l - 1), ps2 = side2(l - 1), ppos = ps1 * ps2;
      const float* const pp = s_patch + poff(l - 1);
      const float* const mp = a.maps + (l - 1) * a.map_stride + c * N * Fp;
      const int o1[2] = {cp_wrap(q1[0] - (l + 1) * g.hi - g.lo, D1), cp_wrap(q1[1] - (l + 1) * g.hi - g.lo, D1)};       // window origins
      const int o2[2] = {cp_wrap(q2[0] - (l + 1) * g.hi2 - g.lo2, D2), cp_wrap(q2[1] - (l + 1) * g.hi2 - g.lo2, D2)};
      const int b1[2] = {cp_wrap(q1[0] - l * g.hi, D1), cp_wrap(q1[1] - l * g.hi, D1)};                                 // origins of the boxes below
      const int b2[2] = {cp_wrap(q2[0] - l * g.hi2, D2), cp_wrap(q2[1] - l * g.hi2, D2)};
      const float inv_sc = 1.f / (float)SC;
      // a thread takes whole window sites (the coordinates and the box test once per site, four 16-byte quads each)
      const int total = NBX * n_win;
      constexpr int SB = 3;            // sites in flight per thread: 12 loads
      for (int i0 = tid; i0 < total; i0 += SB * NT) {
        f32x4 v[SB][4];
        int dsto[SB], pat[SB];
#pragma unroll
        for (int uu = 0; uu < SB; ++uu) {
          const int i = min(i0 + uu * NT, total - 1);
          const int b = i >= n_win, j = i - b * n_win;
          const int wy = cp_div(j, inv_sc), wx = j - wy * SC;
          const int a1 = cp_fold((b ? o1[1] : o1[0]) + wy, D1), a2 = cp_fold((b ? o2[1] : o2[0]) + wx, D2);
          const float* src = mp + (long long)(a1 * D2 + a2) * Fp;
#pragma unroll
          for (int cq = 0; cq < 4; ++cq) v[uu][cq] = *(const f32x4*)(src + (4 * cq < Fp ? 4 * cq : 0));
          pat[uu] = -1;
#pragma unroll
          for (int bb = 0; bb < NBX; ++bb) {
            const int r1 = cp_rel(a1, b1[bb], D1), r2 = cp_rel(a2, b2[bb], D2);
            if (r1 < ps1 && r2 < ps2) pat[uu] = bb * ppos * 16 + (r1 * ps2 + r2) * 16;
          }
          dsto[uu] = b * wstride + j * 16;
        }
#pragma unroll
        for (int uu = 0; uu < SB; ++uu) {
          if (i0 + uu * NT < total) {
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) {
              f32x4 x = v[uu][cq];
              if (pat[uu] >= 0) x = *(const f32x4*)(pp + pat[uu] + 4 * cq);
              f32x4 w;
#pragma unroll
              for (int e = 0; e < 4; ++e) w[e] = 4 * cq + e < F ? cp_pre(pre, x[e]) : 0.f;
              *(f32x4*)(s_win + dsto[uu] + 4 * cq) = w;
            }
          }
        }
      }
      __syncthreads();
      stamp(1);
      if (l == L - 1) {
        if (SUM_AHEAD && tid < 256) {
#pragma unroll
          for (int uu = 0; uu < SU; ++uu) vs[uu] = *(const f32x4*)(lmap + 4 * (long long)min(tid + uu * 256, nq - 1));
        }
        // ... and the sites of this convolution's boxes are marked for the sum (a site of both boxes: either value, they are equal)
        const int bo1[2] = {cp_wrap(q1[0] - L * g.hi, D1), cp_wrap(q1[1] - L * g.hi, D1)};
        const int bo2[2] = {cp_wrap(q2[0] - L * g.hi2, D2), cp_wrap(q2[1] - L * g.hi2, D2)};
        const float inv = 1.f / (float)s2;
        for (int i = tid; i < NBX * s1 * s2; i += NT) {
          const int b = i >= s1 * s2, pos = i - b * s1 * s2, y = cp_div(pos, inv), x = pos - y * s2;
          s_ovl[cp_fold((b ? bo1[1] : bo1[0]) + y, D1) * D2 + cp_fold((b ? bo2[1] : bo2[0]) + x, D2)] = (short)i;
        }
      }
      f32x4 w[T];
#pragma unroll
      for (int t = 0; t < T; ++t) w[t] = *(const f32x4*)(s_wf + ((l - 1) * T + t) * 256 + lane * 4);
      f32x4 bias4;
#pragma unroll
      for (int r = 0; r < 4; ++r) bias4[r] = s_bias[l * 16 + 4 * gq + r];
      const int n_pos = s1 * s2, n_tiles = (n_pos + 15) >> 4;
      const float inv_s2 = 1.f / (float)s2;
      const bool act_out = !g.resnet && l + 1 < L && a.post;
      float* const po = s_patch + poff(l);
      // residual blocks (layers.py:226-228, wavefunctions.py:766-772): an odd convolution ends in selu, an even one adds the
      // block's input h -- the map two convolutions below, with ITS boxes overlaid -- as k_cgen_band's epilogues 11 and 8 do
      const bool res_selu = g.resnet && (l & 1), res_add = g.resnet && !(l & 1);
      int ro1[2] = {0, 0}, ro2[2] = {0, 0}, hb1[2] = {0, 0}, hb2[2] = {0, 0}, hs1 = 1, hs2 = 1;     // (plain networks pay nothing for these)
      const float* hp = s_patch;
      const float* hmap = a.maps;
      if (res_add) {
        for (int bb = 0; bb < NBX; ++bb) {
          ro1[bb] = cp_wrap(q1[bb] - (l + 1) * g.hi, D1); ro2[bb] = cp_wrap(q2[bb] - (l + 1) * g.hi2, D2);      // this convolution's boxes
          hb1[bb] = cp_wrap(q1[bb] - (l - 1) * g.hi, D1); hb2[bb] = cp_wrap(q2[bb] - (l - 1) * g.hi2, D2);      // the boxes of h (convolution l - 2)
        }
        hs1 = side1(l - 2); hs2 = side2(l - 2);
        hp = s_patch + poff(l - 2);
        hmap = a.maps + (l - 2) * a.map_stride + c * N * Fp;
      }
      // the wave's tiles two at a time: two independent accumulator chains keep the matrix pipe busy where one wave per
      // SIMD would wait for every MFMA's result (a tile's own chain, and so its bits, are the same)
      auto locate = [&](int tt, int& b, int& q) {
        b = tt >= n_tiles;
        q = (tt - b * n_tiles) * 16 + p;
        const int qq = q < n_pos ? q : n_pos - 1;
        const int y = cp_div(qq, inv_s2), x = qq - y * s2;
        return s_win + b * wstride + (y * SC + x) * 16 + 4 * gq;
      };
      auto finish = [&](f32x4 acc, int b, int q) {
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc));    // the relu of vmc_act_rt is an asm v_max_f32 (common.hpp: vmc_mfma_settle)
        f32x4 v = acc;
        if (act_out) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = vmc_act_rt(a.act, v[e]);
        } else if (res_selu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = cp_selu(v[e]);
        } else if (res_add) {
          const int qq = q < n_pos ? q : n_pos - 1;
          const int y = cp_div(qq, inv_s2), x = qq - y * s2;
          const int a1 = cp_fold((b ? ro1[1] : ro1[0]) + y, D1), a2 = cp_fold((b ? ro2[1] : ro2[0]) + x, D2);
          f32x4 h = {0.f, 0.f, 0.f, 0.f};
          if (4 * gq < Fp) {
            h = *(const f32x4*)(hmap + (long long)(a1 * D2 + a2) * Fp + 4 * gq);
#pragma unroll
            for (int bb = 0; bb < NBX; ++bb) {
              const int r1 = cp_rel(a1, hb1[bb], D1), r2 = cp_rel(a2, hb2[bb], D2);
              if (r1 < hs1 && r2 < hs2) h = *(const f32x4*)(hp + bb * hs1 * hs2 * 16 + (r1 * hs2 + r2) * 16 + 4 * gq);
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += h[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 4 * gq + e < F ? v[e] : 0.f;
        if (q < n_pos) *(f32x4*)(po + b * n_pos * 16 + q * 16 + 4 * gq) = v;
      };
      int tt = wave;
      for (; NW == 4 && tt + NW < NBX * n_tiles; tt += 2 * NW) {      // (eight waves: two per SIMD take turns already)
        int bA, qA, bB, qB;
        const float* baseA = locate(tt, bA, qA);
        const float* baseB = locate(tt + NW, bB, qB);
        f32x4 accA = bias4, accB = bias4;
        // the operands of tap t + 1 are requested before the MFMAs of tap t (one wave per SIMD: nobody else covers an LDS round trip)
        auto off = [&](int t) { return ((t / KW) * SC + (t % KW)) * 16; };
        f32x4 rA[2], rB[2];
        rA[0] = *(const f32x4*)(baseA + off(0)); rB[0] = *(const f32x4*)(baseB + off(0));
#pragma unroll
        for (int t = 0; t < T; ++t) {
          if (t + 1 < T) { rA[(t + 1) & 1] = *(const f32x4*)(baseA + off(t + 1)); rB[(t + 1) & 1] = *(const f32x4*)(baseB + off(t + 1)); }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            accA = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][e], rA[t & 1][e], accA, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][e], rB[t & 1][e], accB, 0, 0, 0);
          }
        }
        finish(accA, bA, qA);
        finish(accB, bB, qB);
      }
      for (; tt < NBX * n_tiles; tt += NW) {
        int b, q;
        const float* base = locate(tt, b, q);
        f32x4 acc = bias4;
        auto off = [&](int t) { return ((t / KW) * SC + (t % KW)) * 16; };
        f32x4 r[2];
        r[0] = *(const f32x4*)(base + off(0));
#pragma unroll
        for (int t = 0; t < T; ++t) {
          if (t + 1 < T) r[(t + 1) & 1] = *(const f32x4*)(base + off(t + 1));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][e], r[t & 1][e], acc, 0, 0, 0);
        }
        finish(acc, b, q);
      }
    }
    __syncthreads();
    stamp(2);
    // ---- the candidate's logit: the sum of the last map with its two boxes overlaid, in k_cgen_step_tail's order
    {
      const float* const lp = s_patch + poff(L - 1);        // [2][box positions][16]: s_ovl's index is box * positions + position
      const float inv_fq = 1.f / (float)FQ;
      double s = 0.0;
      for (int i0 = tid; i0 < (tid < 256 ? nq : 0); i0 += SU * 256) {       // (the elements of a thread are added in ascending order)
        if (!SUM_AHEAD || i0 != tid) {
#pragma unroll
          for (int uu = 0; uu < SU; ++uu) vs[uu] = *(const f32x4*)(lmap + 4 * (long long)min(i0 + uu * 256, nq - 1));
        }
#pragma unroll
        for (int uu = 0; uu < SU; ++uu) {
          const int i = i0 + uu * 256;
          if (i < nq) {
            const int site = cp_div(i, inv_fq), cq = i - site * FQ;
            const int ov = s_ovl[site];
            f32x4 x = vs[uu];
            if (ov >= 0) x = *(const f32x4*)(lp + ov * 16 + 4 * cq);
            const int c0 = 4 * cq;
#pragma unroll
            for (int e = 0; e < 4; ++e) s += c0 + e < F ? (double)x[e] : 0.0;
          }
        }
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
      if (lane == 0 && wave < 4) s_w[wave] = s;
    }
    if (NW == 8 && !ELOC && tid >= 256 && st + 1 < a.n_steps) {
      // waves 4 .. 7 have no part in the map sum (its order is 256 threads'): they draw the NEXT step's uniforms meanwhile --
      // Philox does not depend on how this step ends, only the spins they multiply do
      const unsigned long long next_step = a.step0 + (unsigned long long)st + 1;
      const uint2 key = make_uint2(a.seed_lo, a.seed_hi);
      const uint32_t gid = (uint32_t)(a.chain_offset + (int)c);
      const int nblk = (N + 3) >> 2;
      for (int b = tid - 256; b < nblk; b += NT - 256) {
        const uint4 r = philox4x32_10(make_uint4((uint32_t)b, gid, (uint32_t)next_step, (uint32_t)(next_step >> 32)), key);
        *(f32x4*)(s_ru + 4 * b) = f32x4{u32_to_uniform(r.x), u32_to_uniform(r.y), u32_to_uniform(r.z), u32_to_uniform(r.w)};
      }
      if (tid == 256) {
        const uint4 ra = philox4x32_10(make_uint4(VMC_ACCEPT_BLOCK, gid, (uint32_t)next_step, (uint32_t)(next_step >> 32)), key);
        s_unext = u32_to_uniform(ra.x);
      }
    }
    __syncthreads();
    stamp(3);
    if (ELOC) {
      if (tid == 0) a.out_sum[st] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);        // k_cgen_rowsum's value
    } else if (tid == 0) {
      const double sd = 0.0 + ((s_w[0] + s_w[1]) + (s_w[2] + s_w[3]));
      const float lnew = (float)sd + 0.f;
      const float uu = s_u;
      const bool acc = vmc_out_accept(a.oact, lnew, cur_logit, uu, 0.5f * __logf(uu));
      if (acc) {                                     // graph_builders.py:67-71
        cur_logit = lnew;
        ++n_acc;
        s_x[dn] += 2.f; s_x[up] -= 2.f;
        a.configs[c * N + dn] = s_x[dn]; a.configs[c * N + up] = s_x[up];
      }
      s_prop[2] = acc ? 1 : 0;
    }
    __syncthreads();
    {                                      // the marks of this step's boxes are taken back
      const int s1 = side1(L - 1), s2 = side2(L - 1);
      const int bo1[2] = {cp_wrap(q1[0] - L * g.hi, D1), cp_wrap(q1[1] - L * g.hi, D1)};
      const int bo2[2] = {cp_wrap(q2[0] - L * g.hi2, D2), cp_wrap(q2[1] - L * g.hi2, D2)};
      const float inv = 1.f / (float)s2;
      for (int i = tid; i < NBX * s1 * s2; i += NT) {
        const int b = i >= s1 * s2, pos = i - b * s1 * s2, y = cp_div(pos, inv), x = pos - y * s2;
        s_ovl[cp_fold((b ? bo1[1] : bo1[0]) + y, D1) * D2 + cp_fold((b ? bo2[1] : bo2[0]) + x, D2)] = -1;
      }
    }
    if (!ELOC && s_prop[2]) {
      // ---- accepted: the boxes become part of the chain's maps
      for (int l = 0; l < L; ++l) {
        const int s1 = side1(l), s2 = side2(l), n_pos = s1 * s2;
        const float* const po = s_patch + poff(l);
        float* const mp = a.maps + l * a.map_stride + c * N * Fp;
        const int o1[2] = {cp_wrap(q1[0] - (l + 1) * g.hi, D1), cp_wrap(q1[1] - (l + 1) * g.hi, D1)};
        const int o2[2] = {cp_wrap(q2[0] - (l + 1) * g.hi2, D2), cp_wrap(q2[1] - (l + 1) * g.hi2, D2)};
        const float inv_fq = 1.f / (float)FQ, inv_s2 = 1.f / (float)s2;
        for (int i = tid; i < 2 * n_pos * FQ; i += NT) {
          const int b = i >= n_pos * FQ, r = i - b * n_pos * FQ, pos = cp_div(r, inv_fq), cq = r - pos * FQ;
          const int y = cp_div(pos, inv_s2), x = pos - y * s2;
          const int a1 = cp_fold((b ? o1[1] : o1[0]) + y, D1), a2 = cp_fold((b ? o2[1] : o2[0]) + x, D2);
          *(f32x4*)(mp + (long long)(a1 * D2 + a2) * Fp + 4 * cq) = *(const f32x4*)(po + b * n_pos * 16 + pos * 16 + 4 * cq);
        }
      }
    }
    stamp(4);
    // ---- the next proposal (k_cgen_step_tail's, k_wide_propose's arithmetic) from the chain as it now stands
    // (all four waves: the rule -- the largest / smallest x u, the lowest index among equal values -- does not depend on the
    // order in which the candidates meet)
    if (!ELOC && st + 1 < a.n_steps) {
      const unsigned long long next_step = a.step0 + (unsigned long long)st + 1;
      const uint2 key = make_uint2(a.seed_lo, a.seed_hi);
      const uint32_t gid = (uint32_t)(a.chain_offset + (int)c);
      float best_hi = -INFINITY, best_lo = INFINITY;
      int idx_hi = 0x7fffffff, idx_lo = 0x7fffffff;
      const int nblk = (N + 3) >> 2;
      for (int b = tid; b < nblk; b += NT) {
        f32x4 ur;
        if (NW == 8) ur = *(const f32x4*)(s_ru + 4 * b);
        else {
          const uint4 r = philox4x32_10(make_uint4((uint32_t)b, gid, (uint32_t)next_step, (uint32_t)(next_step >> 32)), key);
          ur = f32x4{u32_to_uniform(r.x), u32_to_uniform(r.y), u32_to_uniform(r.z), u32_to_uniform(r.w)};
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * b + e;
          if (i < N) {
            const float v = s_x[i] * ur[e];
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
      if (lane == 0) { s_bv[0][wave] = best_hi; s_bi[0][wave] = idx_hi; s_bv[1][wave] = best_lo; s_bi[1][wave] = idx_lo; }
      __syncthreads();
      if (tid == 0) {
        for (int wv = 1; wv < NW; ++wv) {
          const float oh = s_bv[0][wv]; const int ih = s_bi[0][wv];
          if (oh > best_hi || (oh == best_hi && ih < idx_hi)) { best_hi = oh; idx_hi = ih; }
          const float ol = s_bv[1][wv]; const int il = s_bi[1][wv];
          if (ol < best_lo || (ol == best_lo && il < idx_lo)) { best_lo = ol; idx_lo = il; }
        }
        float un = s_unext;
        if (NW != 8) {
          const uint4 ra = philox4x32_10(make_uint4(VMC_ACCEPT_BLOCK, gid, (uint32_t)next_step, (uint32_t)(next_step >> 32)), key);
          un = u32_to_uniform(ra.x);
        }
        s_prop[0] = idx_hi; s_prop[1] = idx_lo; s_u = un;
      }
    }
    __syncthreads();      // the proposal; an accepted move's map writes are behind this for every wave of the workgroup
    stamp(5);
  }
  if (prof) for (int i = 0; i < 6; ++i) a.prof[i] = t_ph[i];
  if (!ELOC && tid == 0) {
    a.logit[c] = cur_logit;
    if (n_acc) atomicAdd(a.accepted, (unsigned long long)n_acc);
  }
}

#undef NBX

template <int K, int KW, int NW, bool ELOC>
hipError_t launch_pw(hipStream_t s, const CgenPatchArgs& a, unsigned grid) {
  const size_t lds = plan_cgen_patch_lds_bytes(a.g);
  hipError_t e = hipFuncSetAttribute((const void*)k_cgen_patch_sweep<K, KW, NW, ELOC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_cgen_patch_sweep<K, KW, NW, ELOC>), dim3(grid), dim3(64 * NW), lds, s, a);
  return hipGetLastError();
}
// eight waves while the fragments of a convolution leave room for two waves per SIMD (measured at 36 x 36 x 16, 5 x 5: 40.2
// against 42.2 ms per sweep on four: profiles/r6_conv_patch_bench.txt), four at 6 x 6 and 7 x 7 taps
template <int K, int KW, bool ELOC>
hipError_t launch_p(hipStream_t s, const CgenPatchArgs& a, unsigned grid) {
  return launch_pw<K, KW, (K * KW <= 25 ? 8 : 4), ELOC>(s, a, grid);
}

}  // namespace

bool cgen_patch_ok(const ConvGeom& g, long long B) { return plan_cgen_patch_ok(g, B); }

hipError_t launch_cgen_patch_sweep(hipStream_t s, const CgenPatchArgs& a) {
  if (a.B <= 0 || a.n_steps <= 0) return hipSuccess;
  if (!cgen_patch_ok(a.g, a.B)) return hipErrorInvalidValue;
  const bool two_d = a.g.KW == a.g.K;
#define CP_CASE(KK) case KK: return two_d ? launch_p<KK, KK, false>(s, a, (unsigned)a.B) : launch_p<KK, 1, false>(s, a, (unsigned)a.B);
  switch (a.g.K) {
    CP_CASE(2) CP_CASE(3) CP_CASE(4) CP_CASE(5) CP_CASE(6) CP_CASE(7)
    default: return hipErrorInvalidValue;
  }
#undef CP_CASE
}

// The sums of the last map of n_rows rows of a row list over the chains whose maps `a.maps` holds: one workgroup per CU walks
// the rows (the LDS of a workgroup is most of a CU's)
hipError_t launch_cgen_patch_rows(hipStream_t s, const CgenPatchArgs& a, int num_cus) {
  if (a.B <= 0 || a.n_rows <= 0) return hipSuccess;
  if (!cgen_patch_ok(a.g, a.B) || !a.rowinfo || !a.bonds || !a.out_sum) return hipErrorInvalidValue;
  const bool two_d = a.g.KW == a.g.K;
  const unsigned grid = (unsigned)(a.n_rows < num_cus ? a.n_rows : num_cus);
#define CP_CASE(KK) case KK: return two_d ? launch_p<KK, KK, true>(s, a, grid) : launch_p<KK, 1, true>(s, a, grid);
  switch (a.g.K) {
    CP_CASE(2) CP_CASE(3) CP_CASE(4) CP_CASE(5) CP_CASE(6) CP_CASE(7)
    default: return hipErrorInvalidValue;
  }
#undef CP_CASE
}
