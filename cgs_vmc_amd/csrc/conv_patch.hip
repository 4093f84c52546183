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
// Shapes: plan_cgen_patch_ok (plan.hpp).
#include "conv.hpp"

#include <type_traits>

namespace {

__device__ __forceinline__ float cp_pre(int pre, float x) { return pre < 0 ? x : vmc_act_rt(pre, x); }
__device__ __forceinline__ int cp_wrap(int v, int d) { v %= d; return v < 0 ? v + d : v; }

template <int K, int KW>
__global__ __launch_bounds__(256, 1) void k_cgen_patch_sweep(CgenPatchArgs a) {
  constexpr int T = K * KW, NF0 = (T + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ double s_w[4];
  __shared__ int s_prop[3];                    // the site raised (iup), the site lowered (idn), accepted
  __shared__ float s_u;
  const ConvGeom g = a.g;
  const int D1 = g.D1, D2 = g.D2, N = g.N, F = g.F, Fp = a.Fp, L = g.n_conv, FQ = Fp >> 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, gq = lane >> 4;
  const long long c = blockIdx.x;
  auto side1 = [&](int l) { return (l + 1) * (K - 1) + 1; };
  auto side2 = [&](int l) { return (l + 1) * (KW - 1) + 1; };
  auto poff = [&](int l) { int o = 0; for (int j = 0; j < l; ++j) o += 2 * side1(j) * side2(j) * 16; return o; };
  // ---- LDS (plan_cgen_patch_lds_bytes): spins, weight fragments, biases, the boxes of every convolution, two windows
  float* const s_x = sm;
  float* const s_wf = s_x + ((N + 3) & ~3);
  float* const s_bias = s_wf + (L - 1) * T * 256;
  float* const s_patch = s_bias + L * 16;
  float* const s_win = s_patch + poff(L);
  int wstride = (2 * K - 1) * (2 * KW - 1);
  for (int l = 1; l < L; ++l) wstride = max(wstride, (side1(l) + K - 1) * (side2(l) + KW - 1) * 16);

  // ---- once per launch: the chain, the parameters
  const long long w_base1 = (long long)T * F + F, w_per = (long long)T * F * F + F;     // theta: w_0, b_0, then (w_l, b_l)
  for (int i = tid; i < N; i += 256) s_x[i] = a.configs[c * N + i];
  for (int i = tid; i < (L - 1) * T * 256; i += 256) {
    // fragment (l, tap t) of lane ln, MFMA e: the A operand of k_cgen_band -- output channel ln & 15 against input channel 4 (ln >> 4) + e
    const int l1 = i / (T * 256), r = i - l1 * T * 256, t = r >> 8, ln = (r >> 2) & 63, e = r & 3;
    const int c_in = 4 * (ln >> 4) + e, fo = ln & 15;
    const float* w = a.theta + w_base1 + l1 * w_per;
    s_wf[i] = (c_in < F && fo < F) ? w[((long long)t * F + c_in) * F + fo] : 0.f;
  }
  for (int i = tid; i < L * 16; i += 256) {
    const int l = i >> 4, f = i & 15;
    const float* b = l == 0 ? a.theta + (long long)T * F : a.theta + w_base1 + (l - 1) * w_per + (long long)T * F * F;
    s_bias[i] = f < F ? b[f] : 0.f;
  }
  float wf0[NF0];                              // first convolution: the taps run over the MFMA's k index (k_cgen_band<FIRST>)
  int toff0[NF0];
#pragma unroll
  for (int m = 0; m < NF0; ++m) {
    const int t = 4 * m + gq, tt = t < T ? t : 0;
    wf0[m] = (t < T && p < F) ? a.theta[(long long)t * F + p] : 0.f;
    toff0[m] = (tt / KW) * (2 * KW - 1) + (tt % KW);
  }
  float cur_logit = 0.f;
  unsigned n_acc = 0;
  if (tid == 0) { s_prop[0] = a.iup[c]; s_prop[1] = a.idn[c]; s_u = a.u[c]; cur_logit = a.logit[c]; }
  __syncthreads();

  const int pre = a.post ? -1 : a.act;
  for (long long st = 0; st < a.n_steps; ++st) {
    const int up = s_prop[0], dn = s_prop[1];
    const int q1[2] = {up / D2, dn / D2};
    const int q2[2] = {up - q1[0] * D2, dn - q1[1] * D2};
    // ---- convolution 0: the windows are the candidate's spins (the exchanged pair negated)
    {
      constexpr int SR = 2 * K - 1, SC = 2 * KW - 1, n_win = SR * SC, n_pos = K * KW, n_tiles = (n_pos + 15) >> 4;
      for (int i = tid; i < 2 * n_win; i += 256) {
        const int b = i >= n_win, j = i - b * n_win, wy = j / SC, wx = j - wy * SC;
        const int site = cp_wrap(q1[b] - g.hi - g.lo + wy, D1) * D2 + cp_wrap(q2[b] - g.hi2 - g.lo2 + wx, D2);
        const float x = s_x[site];
        s_win[b * wstride + j] = (site == up || site == dn) ? -x : x;
      }
      __syncthreads();
      f32x4 bias4;
#pragma unroll
      for (int r = 0; r < 4; ++r) bias4[r] = s_bias[4 * gq + r];
      for (int tt = wave; tt < 2 * n_tiles; tt += 4) {
        const int b = tt >= n_tiles, tile = tt - b * n_tiles;
        const int q = tile * 16 + p, qq = q < n_pos ? q : n_pos - 1;
        const int y = qq / KW, x = qq - y * KW;
        f32x4 acc = bias4;
        const float* base = s_win + b * wstride + y * SC + x;
#pragma unroll
        for (int m = 0; m < NF0; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf0[m], base[toff0[m]], acc, 0, 0, 0);
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc));    // the relu of vmc_act_rt is an asm v_max_f32 (common.hpp: vmc_mfma_settle)
        f32x4 v = acc;
        if (a.post) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = vmc_act_rt(a.act, v[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 4 * gq + e < F ? v[e] : 0.f;
        if (q < n_pos) *(f32x4*)(s_patch + b * n_pos * 16 + q * 16 + 4 * gq) = v;
      }
    }
    // ---- the convolutions behind it
    for (int l = 1; l < L; ++l) {
      __syncthreads();                 // the boxes of convolution l - 1 are written; the windows' readers are done
      const int s1 = side1(l), s2 = side2(l), SC = s2 + KW - 1, n_win = (s1 + K - 1) * SC;
      const int ps1 = side1(l - 1), ps2 = side2(l - 1), ppos = ps1 * ps2;
      const float* const pp = s_patch + poff(l - 1);
      const float* const mp = a.maps + (l - 1) * a.map_stride + c * N * Fp;
      const int o1[2] = {q1[0] - (l + 1) * g.hi - g.lo, q1[1] - (l + 1) * g.hi - g.lo};        // window origins
      const int o2[2] = {q2[0] - (l + 1) * g.hi2 - g.lo2, q2[1] - (l + 1) * g.hi2 - g.lo2};
      const int b1[2] = {q1[0] - l * g.hi, q1[1] - l * g.hi};                                  // origins of the boxes below
      const int b2[2] = {q2[0] - l * g.hi2, q2[1] - l * g.hi2};
      const int total = 2 * n_win * 4;
      constexpr int SB = 4;            // loads in flight per thread
      for (int i0 = tid; i0 < total; i0 += SB * 256) {
        f32x4 v[SB];
        int dsto[SB], cqs[SB], pat[SB];
#pragma unroll
        for (int uu = 0; uu < SB; ++uu) {
          const int i = min(i0 + uu * 256, total - 1);
          const int b = i >= n_win * 4, r = i - b * n_win * 4, j = r >> 2, cq = r & 3;
          const int wy = j / SC, wx = j - wy * SC;
          const int a1 = cp_wrap(o1[b] + wy, D1), a2 = cp_wrap(o2[b] + wx, D2);
          v[uu] = *(const f32x4*)(mp + (long long)(a1 * D2 + a2) * Fp + (4 * cq < Fp ? 4 * cq : 0));
          pat[uu] = -1;
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) {
            const int r1 = cp_wrap(a1 - b1[bb], D1), r2 = cp_wrap(a2 - b2[bb], D2);
            if (r1 < ps1 && r2 < ps2) pat[uu] = bb * ppos * 16 + (r1 * ps2 + r2) * 16 + 4 * cq;
          }
          dsto[uu] = b * wstride + j * 16 + 4 * cq;
          cqs[uu] = 4 * cq;
        }
#pragma unroll
        for (int uu = 0; uu < SB; ++uu) {
          if (i0 + uu * 256 < total) {
            f32x4 x = v[uu];
            if (pat[uu] >= 0) x = *(const f32x4*)(pp + pat[uu]);
            f32x4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = cqs[uu] + e < F ? cp_pre(pre, x[e]) : 0.f;
            *(f32x4*)(s_win + dsto[uu]) = w;
          }
        }
      }
      __syncthreads();
      f32x4 w[T];
#pragma unroll
      for (int t = 0; t < T; ++t) w[t] = *(const f32x4*)(s_wf + ((l - 1) * T + t) * 256 + lane * 4);
      f32x4 bias4;
#pragma unroll
      for (int r = 0; r < 4; ++r) bias4[r] = s_bias[l * 16 + 4 * gq + r];
      const int n_pos = s1 * s2, n_tiles = (n_pos + 15) >> 4;
      const bool act_out = l + 1 < L && a.post;
      float* const po = s_patch + poff(l);
      for (int tt = wave; tt < 2 * n_tiles; tt += 4) {
        const int b = tt >= n_tiles, tile = tt - b * n_tiles;
        const int q = tile * 16 + p, qq = q < n_pos ? q : n_pos - 1;
        const int y = qq / s2, x = qq - y * s2;
        f32x4 acc = bias4;
        const float* base = s_win + b * wstride + (y * SC + x) * 16 + 4 * gq;
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const f32x4 bv = *(const f32x4*)(base + ((t / KW) * SC + (t % KW)) * 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][e], bv[e], acc, 0, 0, 0);
        }
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc));
        f32x4 v = acc;
        if (act_out) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = vmc_act_rt(a.act, v[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 4 * gq + e < F ? v[e] : 0.f;
        if (q < n_pos) *(f32x4*)(po + b * n_pos * 16 + q * 16 + 4 * gq) = v;
      }
    }
    __syncthreads();
    // ---- the candidate's logit: the sum of the last map with its two boxes overlaid, in k_cgen_step_tail's order
    {
      const int ls1 = side1(L - 1), ls2 = side2(L - 1), lpos = ls1 * ls2;
      const float* const lp = s_patch + poff(L - 1);
      const float* const mp = a.maps + (L - 1) * a.map_stride + c * N * Fp;
      const int b1[2] = {q1[0] - L * g.hi, q1[1] - L * g.hi};
      const int b2[2] = {q2[0] - L * g.hi2, q2[1] - L * g.hi2};
      const int nq = N * FQ;
      double s = 0.0;
      for (int i = tid; i < nq; i += 256) {
        const int site = i / FQ, cq = i - site * FQ;
        f32x4 v = *(const f32x4*)(mp + 4 * (long long)i);
        const int a1 = site / D2, a2 = site - a1 * D2;
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          const int r1 = cp_wrap(a1 - b1[bb], D1), r2 = cp_wrap(a2 - b2[bb], D2);
          if (r1 < ls1 && r2 < ls2) v = *(const f32x4*)(lp + bb * lpos * 16 + (r1 * ls2 + r2) * 16 + 4 * cq);
        }
        const int c0 = 4 * cq;
#pragma unroll
        for (int e = 0; e < 4; ++e) s += c0 + e < F ? (double)v[e] : 0.0;
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
      if (lane == 0) s_w[wave] = s;
    }
    __syncthreads();
    if (tid == 0) {
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
    if (s_prop[2]) {
      // ---- accepted: the boxes become part of the chain's maps
      for (int l = 0; l < L; ++l) {
        const int s1 = side1(l), s2 = side2(l), n_pos = s1 * s2;
        const float* const po = s_patch + poff(l);
        float* const mp = a.maps + l * a.map_stride + c * N * Fp;
        for (int i = tid; i < 2 * n_pos * FQ; i += 256) {
          const int b = i >= n_pos * FQ, r = i - b * n_pos * FQ, pos = r / FQ, cq = r - pos * FQ;
          const int y = pos / s2, x = pos - y * s2;
          const int a1 = cp_wrap(q1[b] - (l + 1) * g.hi + y, D1), a2 = cp_wrap(q2[b] - (l + 1) * g.hi2 + x, D2);
          *(f32x4*)(mp + (long long)(a1 * D2 + a2) * Fp + 4 * cq) = *(const f32x4*)(po + b * n_pos * 16 + pos * 16 + 4 * cq);
        }
      }
    }
    // ---- the next proposal (k_cgen_step_tail's, k_wide_propose's arithmetic) from the chain as it now stands
    if (st + 1 < a.n_steps && tid < 64) {
      const unsigned long long next_step = a.step0 + (unsigned long long)st + 1;
      const uint2 key = make_uint2(a.seed_lo, a.seed_hi);
      const uint32_t gid = (uint32_t)(a.chain_offset + (int)c);
      float best_hi = -INFINITY, best_lo = INFINITY;
      int idx_hi = 0x7fffffff, idx_lo = 0x7fffffff;
      const int nblk = (N + 3) >> 2;
      for (int b = lane; b < nblk; b += 64) {
        const uint4 r = philox4x32_10(make_uint4((uint32_t)b, gid, (uint32_t)next_step, (uint32_t)(next_step >> 32)), key);
        const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * b + e;
          if (i < N) {
            const float v = s_x[i] * u32_to_uniform(rr[e]);
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
      if (lane == 0) {
        const uint4 ra = philox4x32_10(make_uint4(VMC_ACCEPT_BLOCK, gid, (uint32_t)next_step, (uint32_t)(next_step >> 32)), key);
        s_prop[0] = idx_hi; s_prop[1] = idx_lo; s_u = u32_to_uniform(ra.x);
      }
    }
    __syncthreads();      // the proposal; an accepted move's map writes are behind this for every wave of the workgroup
  }
  if (tid == 0) {
    a.logit[c] = cur_logit;
    if (n_acc) atomicAdd(a.accepted, (unsigned long long)n_acc);
  }
}

template <int K, int KW>
hipError_t launch_p(hipStream_t s, const CgenPatchArgs& a) {
  const size_t lds = plan_cgen_patch_lds_bytes(a.g);
  hipError_t e = hipFuncSetAttribute((const void*)k_cgen_patch_sweep<K, KW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_cgen_patch_sweep<K, KW>), dim3((unsigned)a.B), dim3(256), lds, s, a);
  return hipGetLastError();
}

}  // namespace

bool cgen_patch_ok(const ConvGeom& g, long long B) { return plan_cgen_patch_ok(g, B); }

hipError_t launch_cgen_patch_sweep(hipStream_t s, const CgenPatchArgs& a) {
  if (a.B <= 0 || a.n_steps <= 0) return hipSuccess;
  if (!cgen_patch_ok(a.g, a.B)) return hipErrorInvalidValue;
  const bool two_d = a.g.KW == a.g.K;
#define CP_CASE(KK) case KK: return two_d ? launch_p<KK, KK>(s, a) : launch_p<KK, 1>(s, a);
  switch (a.g.K) {
    CP_CASE(2) CP_CASE(3) CP_CASE(4) CP_CASE(5) CP_CASE(6) CP_CASE(7)
    default: return hipErrorInvalidValue;
  }
#undef CP_CASE
}
