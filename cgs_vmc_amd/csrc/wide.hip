// fully_connected / rbm with fc_layer_size > 256 (utils.py:105 allows any size): the general path.
// The register-resident kernels (k_tail16 / k_sweep16 / k_backprop16) hold all hidden units of a
// row tile in registers, which ends at 256 units.  Wider layers go through materialised rows and
// the library's LDS-tiled fp32-MFMA GEMM (grad.hip): the rank-2 first layer of every connected /
// proposed configuration is written out as a row of activations, the H x H layers are GEMMs with
// the activation epilogue, and the Metropolis step is a short sequence of launches per mc_step
// (graph_builders.py:38-89) instead of one persistent kernel.  Same arithmetic, same Philox
// streams, same accept rule; throughput is that of the generic GEMM.
#include "common.hpp"

namespace {

__device__ __forceinline__ float wave_sum_w(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// a[r][:] = f(z1[chain] + coef (W1[i] - W1[j])) for the rows {chain, +-(bond+1) or 0} of a row
// list (coef = -2 s_i: the exchange of the bond's two antiparallel spins, operators.py:162-163)
__global__ void k_wide_rows_act(const float* __restrict__ z1, const float* __restrict__ w1p,
                                const int2* __restrict__ rowinfo, const int2* __restrict__ bonds,
                                long long row0, int n_rows, int Hp, int act, float* __restrict__ out) {
  const int q = Hp >> 2;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < (long long)n_rows * q;
       idx += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(idx / q), c4 = (int)(idx % q) * 4;
    const int2 ri = rowinfo[row0 + r];
    const int bs = ri.y;
    f32x4 z = *(const f32x4*)(z1 + (long long)ri.x * Hp + c4);
    if (bs != 0) {
      const int2 ab = bonds[(bs > 0 ? bs : -bs) - 1];
      const float coef = bs > 0 ? -2.f : 2.f;
      const f32x4 x = *(const f32x4*)(w1p + (long long)ab.x * Hp + c4);
      const f32x4 y = *(const f32x4*)(w1p + (long long)ab.y * Hp + c4);
#pragma unroll
      for (int e = 0; e < 4; ++e) z[e] = fmaf(coef, x[e] - y[e], z[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) z[e] = vmc_act_rt(act, z[e]);
    *(f32x4*)(out + (long long)r * Hp + c4) = z;
  }
}

// row r's logit from its output dot `sd` (+ b_out, + the RBM on-site term), stored as the logit or as the ratio term
__device__ __forceinline__ void wide_out_finish(double sd, int r, const float* __restrict__ bout,
                                                const int2* __restrict__ rowinfo, long long row0,
                                                const float* __restrict__ half_jx,
                                                const float* __restrict__ logit_base, int oact, int ratio,
                                                float* __restrict__ out, const WideOnsite& on) {
  float logit = (float)sd + bout[0];
  if (on.base) {   // RestrictedBoltzmannNetwork: + x' . w_on (wavefunctions.py:436), rank-2 in the exchange
    const int2 ri = rowinfo[row0 + r];
    float o = on.base[ri.x];
    if (ri.y != 0) {
      const int2 ab = on.bonds[(ri.y > 0 ? ri.y : -ri.y) - 1];
      o = fmaf(ri.y > 0 ? -2.f : 2.f, on.won[ab.x] - on.won[ab.y], o);
    }
    if (on.iup) o = fmaf(2.f, on.won[on.idn[r]] - on.won[on.iup[r]], o);
    logit += o;
  }
  if (ratio) {
    const int2 ri = rowinfo[row0 + r];
    const int bs = ri.y;
    out[row0 + r] = half_jx[(bs > 0 ? bs : -bs) - 1] * vmc_out_ratio(oact, logit, logit_base[ri.x]);
  } else {
    out[row0 + r] = logit;
  }
}

// logit[r] = a[r] . w_out + b_out (one wave per row); ratio: out = 0.5 jx[bond] psi'/psi
__global__ __launch_bounds__(256) void k_wide_out(const float* __restrict__ a, const float* __restrict__ wout,
                                                  const float* __restrict__ bout, int n_rows, int H, int Hp,
                                                  const int2* __restrict__ rowinfo, long long row0,
                                                  const float* __restrict__ half_jx,
                                                  const float* __restrict__ logit_base, int oact, int ratio,
                                                  float* __restrict__ out, WideOnsite on) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  // the dot runs over up to 4096 terms of one sign (RBM: log cosh values, logit ~ H / 2): summed in
  // double so that logit' - logit keeps its digits (as k_tail0 does); the products are exact in double
  double sd = 0.0;
  for (int h = lane; h < H; h += 64) sd += (double)a[(long long)r * Hp + h] * (double)wout[h];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) sd += __shfl_xor(sd, m);
  if (lane == 0) wide_out_finish(sd, r, bout, rowinfo, row0, half_jx, logit_base, oact, ratio, out, on);
}

// the same from the row-dot partials of the last H x H layer (GemmArgs epilogue 10, grad.hip): one thread per row
// adds the column tiles' partials in ascending order
__global__ __launch_bounds__(256) void k_wide_out_part(const double* __restrict__ part, int n_part,
                                                       const float* __restrict__ bout, int n_rows,
                                                       const int2* __restrict__ rowinfo, long long row0,
                                                       const float* __restrict__ half_jx,
                                                       const float* __restrict__ logit_base, int oact, int ratio,
                                                       float* __restrict__ out, WideOnsite on) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rows) return;
  double sd = 0.0;
  for (int t = 0; t < n_part; ++t) sd += part[(long long)t * n_rows + r];
  wide_out_finish(sd, r, bout, rowinfo, row0, half_jx, logit_base, oact, ratio, out, on);
}

// proposals of one mc_step (graph_builders.py:59-65): one wave per chain
__global__ __launch_bounds__(256) void k_wide_propose(const float* __restrict__ configs, int B, int N,
                                                      uint32_t seed_lo, uint32_t seed_hi, int chain_offset,
                                                      unsigned long long step, const int* __restrict__ inj_up,
                                                      const int* __restrict__ inj_dn,
                                                      const float* __restrict__ inj_u, int* __restrict__ iup,
                                                      int* __restrict__ idn, float* __restrict__ u) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= B) return;
  if (inj_up) {
    if (lane == 0) { iup[c] = inj_up[c]; idn[c] = inj_dn[c]; u[c] = inj_u[c]; }
    return;
  }
  const float* x = configs + (long long)c * N;
  const uint2 key = make_uint2(seed_lo, seed_hi);
  const uint32_t gid = (uint32_t)(chain_offset + c);
  float best_hi = -INFINITY, best_lo = INFINITY;
  int idx_hi = 0x7fffffff, idx_lo = 0x7fffffff;
  const int nblk = (N + 3) >> 2;
  for (int b = lane; b < nblk; b += 64) {
    const uint4 r = philox4x32_10(make_uint4((uint32_t)b, gid, (uint32_t)step, (uint32_t)(step >> 32)), key);
    const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = 4 * b + e;
      if (i < N) {
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
  if (lane == 0) {
    const uint4 ra = philox4x32_10(make_uint4(VMC_ACCEPT_BLOCK, gid, (uint32_t)step, (uint32_t)(step >> 32)), key);
    iup[c] = idx_hi; idn[c] = idx_lo; u[c] = u32_to_uniform(ra.x);
  }
}

// One launch per mc_step of the general sampler (graph_builders.py:38-89), one wave per chain, sixteen chains
// per workgroup.  In program order:
//  (1) Metropolis test of the PREVIOUS step's proposal (`do_accept`): logit' = a_last[chain] . w_out + b_out
//      (+ the RBM on-site term) in the arithmetic of k_wide_out -- double accumulation, lane h % 64 -- and the
//      accept rule of graph_builders.py:75-88;
//  (2) this step's proposal (`do_propose`, graph_builders.py:59-65) from the chain as it stands AFTER (1): the
//      flipped spins are patched into the loaded values, the stores to `configs` are issued behind the loads;
//  (3) one pass over the chain's first-layer row: z1 += 2 (W1[i_dn] - W1[i_up]) of the accepted move (the
//      same fmaf the candidate was built with: no second copy of the candidate rows is kept), then the new
//      candidate's activations a0 = f(z1 + 2 (W1[i_dn'] - W1[i_up'])) for the GEMMs that follow.
// `a_last` and `a0` may be the same buffer (an even number of H x H layers): a wave reads its row of a_last
// in (1), whose result (3) depends on, and writes the same row.  One atomic per workgroup carries the accept
// count (one per accepted chain serialised ~2000 atomics on one address: 54 us per step at 4096 chains).
// Round 5: was four launches (propose, build, out, accept: 33 us + four launch gaps per step at 4096 x 1024).
#define WIDE_STEP_CHAINS 16
__global__ __launch_bounds__(64 * WIDE_STEP_CHAINS) void k_wide_step(WideStepArgs s) {
  __shared__ int s_acc[WIDE_STEP_CHAINS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * WIDE_STEP_CHAINS + w;
  const bool live = c < s.B;
  bool acc = false;
  int pu = 0, pd = 0;
  float lnew = 0.f, onew = 0.f;
  if (live && s.do_accept) {
    pu = s.iup[c]; pd = s.idn[c];
    double sd = 0.0;
    if (s.dot_part) {                                 // the last layer's GEMM left the dot as column-tile partials
      for (int t = 0; t < s.n_part; ++t) sd += s.dot_part[(long long)t * s.B + c];
    } else {
      const float* a = s.a_last + (long long)c * s.Hp;
      for (int h0 = lane; h0 < s.H; h0 += 512) {      // the order of k_wide_out (h = lane, lane + 64, ...), eight loads in flight
        float av[8], wv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int h = min(h0 + 64 * i, s.H - 1); av[i] = a[h]; wv[i] = s.wout[h]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) if (h0 + 64 * i < s.H) sd += (double)av[i] * (double)wv[i];
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) sd += __shfl_xor(sd, m);
    }
    lnew = (float)sd + s.bout[0];
    if (s.onsite) {   // RestrictedBoltzmannNetwork: + x' . w_on (wavefunctions.py:436), rank-2 in the exchange
      onew = fmaf(2.f, s.won[pd] - s.won[pu], s.onsite[c]);
      lnew += onew;
    }
    const float uu = s.u[c];
    acc = vmc_out_accept(s.oact, lnew, s.logit[c], uu, 0.5f * __logf(uu));
  }
  acc = __builtin_amdgcn_readfirstlane(acc ? 1 : 0) != 0;
  int nu = 0, nd = 0;
  float nuu = 0.f;
  if (live && s.do_propose) {
    if (s.inj_up) {
      nu = s.inj_up[c]; nd = s.inj_dn[c]; nuu = s.inj_u[c];
    } else {
      const float* x = s.configs + (long long)c * s.N;
      const uint2 key = make_uint2(s.seed_lo, s.seed_hi);
      const uint32_t gid = (uint32_t)(s.chain_offset + c);
      float best_hi = -INFINITY, best_lo = INFINITY;
      int idx_hi = 0x7fffffff, idx_lo = 0x7fffffff;
      const int nblk = (s.N + 3) >> 2;
      for (int b = lane; b < nblk; b += 64) {
        const uint4 r = philox4x32_10(make_uint4((uint32_t)b, gid, (uint32_t)s.step, (uint32_t)(s.step >> 32)), key);
        const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * b + e;
          if (i < s.N) {
            float xv = x[i];
            if (acc) xv += i == pd ? 2.f : (i == pu ? -2.f : 0.f);   // graph_builders.py:67-71, not stored yet
            const float v = xv * u32_to_uniform(rr[e]);
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
      const uint4 ra = philox4x32_10(make_uint4(VMC_ACCEPT_BLOCK, gid, (uint32_t)s.step, (uint32_t)(s.step >> 32)), key);
      nu = idx_hi; nd = idx_lo; nuu = u32_to_uniform(ra.x);
    }
    nu = __builtin_amdgcn_readfirstlane(nu); nd = __builtin_amdgcn_readfirstlane(nd);
  }
  if (live && lane == 0) {          // (behind every load of this chain's spins and of its previous proposal)
    if (acc) {
      s.configs[(long long)c * s.N + pd] += 2.f;
      s.configs[(long long)c * s.N + pu] -= 2.f;
      s.logit[c] = lnew;
      if (s.onsite) s.onsite[c] = onew;
    }
    if (s.do_accept && s.acc_mask) s.acc_mask[c] = acc ? 1 : 0;
    if (s.do_propose) { s.iup[c] = nu; s.idn[c] = nd; s.u[c] = nuu; }
  }
  if (lane == 0) s_acc[w] = acc ? 1 : 0;
  if (live && (acc || s.do_propose)) {
    float* z1 = s.z1 + (long long)c * s.Hp;
    float* a0 = s.a0 + (long long)c * s.Hp;
    const float* xo = s.w1p + (long long)pd * s.Hp; const float* yo = s.w1p + (long long)pu * s.Hp;
    const float* xn = s.w1p + (long long)nd * s.Hp; const float* yn = s.w1p + (long long)nu * s.Hp;
    // 1024 units at a time: every load of a block is issued before its first store (the stores may alias the
    // loads as far as the compiler knows: it would serialise a load -- store -- load chain per 16 bytes)
    for (int base = 4 * lane; base < s.Hp; base += 1024) {
      f32x4 z[4], x[4], y[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c4 = min(base + 256 * i, s.Hp - 4);               // (clamped: unconditional loads)
        z[i] = *(const f32x4*)(z1 + c4);
        if (acc) { x[i] = *(const f32x4*)(xo + c4); y[i] = *(const f32x4*)(yo + c4); }
      }
      if (acc) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) z[i][e] = fmaf(2.f, x[i][e] - y[i][e], z[i][e]);
      }
      f32x4 a[4];
      if (s.do_propose) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c4 = min(base + 256 * i, s.Hp - 4);
          x[i] = *(const f32x4*)(xn + c4); y[i] = *(const f32x4*)(yn + c4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) a[i][e] = vmc_act_rt(s.act, fmaf(2.f, x[i][e] - y[i][e], z[i][e]));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c4 = base + 256 * i;
        if (c4 < s.Hp) {
          if (acc) *(f32x4*)(z1 + c4) = z[i];
          if (s.do_propose) *(f32x4*)(a0 + c4) = a[i];
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && s.do_accept) {
    int n = 0;
#pragma unroll
    for (int i = 0; i < WIDE_STEP_CHAINS; ++i) n += s_acc[i];
    if (n) atomicAdd(s.accepted, (unsigned long long)n);
  }
}

// delta of the last hidden layer: d logit / d z_L = w_out (.) f'(z_L) (x the output-activation factor)
// (`dact` != nullptr: the stored f'(z) of the layer -- the cosine, whose derivative is no function of a)
__global__ void k_wide_delta_last(const float* __restrict__ a_last, const float* __restrict__ wout,
                                  const float* __restrict__ oscale, int B, int H, int Hp, int act,
                                  float* __restrict__ delta, const float* __restrict__ dact) {
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < (long long)B * Hp;
       idx += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(idx / Hp), h = (int)(idx % Hp);
    const float a = a_last[idx];
    const float d = dact ? dact[idx] : vmc_dact_rt(act, a, a);
    delta[idx] = h < H ? wout[h] * d * (oscale ? oscale[b] : 1.f) : 0.f;
  }
}

int blocks_for(long long n) { const long long b = (n + 255) / 256; return (int)(b < 8192 ? (b < 1 ? 1 : b) : 8192); }

}  // namespace

hipError_t launch_wide_rows_act(hipStream_t s, const float* z1, const float* w1p, const int2* rowinfo,
                                const int2* bonds, long long row0, int n_rows, int Hp, int act, float* out) {
  if (n_rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_wide_rows_act, dim3(blocks_for((long long)n_rows * (Hp / 4))), dim3(256), 0, s, z1, w1p,
                     rowinfo, bonds, row0, n_rows, Hp, act, out);
  return hipGetLastError();
}

hipError_t launch_wide_out(hipStream_t s, const float* a, const float* wout, const float* bout, int n_rows,
                           int H, int Hp, const int2* rowinfo, long long row0, const float* half_jx,
                           const float* logit_base, int oact, bool ratio, float* out, const WideOnsite& on) {
  if (n_rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_wide_out, dim3((n_rows + 3) / 4), dim3(256), 0, s, a, wout, bout, n_rows, H, Hp, rowinfo,
                     row0, half_jx, logit_base, oact, ratio ? 1 : 0, out, on);
  return hipGetLastError();
}

hipError_t launch_wide_out_part(hipStream_t s, const double* part, int n_part, const float* bout, int n_rows,
                                const int2* rowinfo, long long row0, const float* half_jx, const float* logit_base,
                                int oact, bool ratio, float* out, const WideOnsite& on) {
  if (n_rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_wide_out_part, dim3((n_rows + 255) / 256), dim3(256), 0, s, part, n_part, bout, n_rows, rowinfo,
                     row0, half_jx, logit_base, oact, ratio ? 1 : 0, out, on);
  return hipGetLastError();
}

hipError_t launch_wide_propose(hipStream_t s, const float* configs, int B, int N, uint32_t seed_lo,
                               uint32_t seed_hi, int chain_offset, unsigned long long step, const int* inj_up,
                               const int* inj_dn, const float* inj_u, int* iup, int* idn, float* u) {
  hipLaunchKernelGGL(k_wide_propose, dim3((B + 3) / 4), dim3(256), 0, s, configs, B, N, seed_lo, seed_hi,
                     chain_offset, step, inj_up, inj_dn, inj_u, iup, idn, u);
  return hipGetLastError();
}

hipError_t launch_wide_step(hipStream_t st, const WideStepArgs& s) {
  if (s.B <= 0 || (!s.do_accept && !s.do_propose)) return hipSuccess;
  hipLaunchKernelGGL(k_wide_step, dim3((s.B + WIDE_STEP_CHAINS - 1) / WIDE_STEP_CHAINS), dim3(64 * WIDE_STEP_CHAINS), 0, st, s);
  return hipGetLastError();
}

hipError_t launch_wide_delta_last(hipStream_t s, const float* a_last, const float* wout, const float* oscale,
                                  int B, int H, int Hp, int act, float* delta, const float* dact) {
  hipLaunchKernelGGL(k_wide_delta_last, dim3(blocks_for((long long)B * Hp)), dim3(256), 0, s, a_last, wout,
                     oscale, B, H, Hp, act, delta, dact);
  return hipGetLastError();
}
