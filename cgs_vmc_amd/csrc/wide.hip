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
  const float s = (float)sd;
  if (lane == 0) {
    float logit = s + bout[0];
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

// candidate first layer of every chain: zc = z1 + 2 (W1[i_dn] - W1[i_up]), a0 = f(zc)
__global__ void k_wide_build(const float* __restrict__ z1, const float* __restrict__ w1p,
                             const int* __restrict__ iup, const int* __restrict__ idn, int B, int Hp, int act,
                             float* __restrict__ zc, float* __restrict__ a0) {
  const int q = Hp >> 2;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < (long long)B * q;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx / q), c4 = (int)(idx % q) * 4;
    f32x4 z = *(const f32x4*)(z1 + (long long)c * Hp + c4);
    const f32x4 x = *(const f32x4*)(w1p + (long long)idn[c] * Hp + c4);
    const f32x4 y = *(const f32x4*)(w1p + (long long)iup[c] * Hp + c4);
    f32x4 a;
#pragma unroll
    for (int e = 0; e < 4; ++e) { z[e] = fmaf(2.f, x[e] - y[e], z[e]); a[e] = vmc_act_rt(act, z[e]); }
    *(f32x4*)(zc + (long long)c * Hp + c4) = z;
    *(f32x4*)(a0 + (long long)c * Hp + c4) = a;
  }
}

// Metropolis test and commit (graph_builders.py:75-88).  A workgroup takes 16 chains: 16 threads decide and
// commit the scalars, one atomic per workgroup carries the accept count (one per accepted chain serialised
// ~2000 atomics on one address: 54 us per step at 4096 chains), then all threads copy the accepted chains'
// candidate first layer in 16-byte pieces.
#define WIDE_ACC_CHAINS 16
__global__ __launch_bounds__(256) void k_wide_accept(float* __restrict__ configs, float* __restrict__ z1,
                                                     const float* __restrict__ zc, float* __restrict__ logit,
                                                     const float* __restrict__ lnew, const int* __restrict__ iup,
                                                     const int* __restrict__ idn, const float* __restrict__ u,
                                                     int B, int N, int Hp, int oact,
                                                     unsigned long long* __restrict__ accepted,
                                                     unsigned char* __restrict__ acc_mask,
                                                     float* __restrict__ onsite, const float* __restrict__ won) {
  __shared__ int s_acc[WIDE_ACC_CHAINS];
  const int c0 = blockIdx.x * WIDE_ACC_CHAINS;
  if (threadIdx.x < WIDE_ACC_CHAINS) {
    const int c = c0 + threadIdx.x;
    bool acc = false;
    if (c < B) {
      const float uu = u[c];
      acc = vmc_out_accept(oact, lnew[c], logit[c], uu, 0.5f * __logf(uu));
      if (acc) {
        configs[(long long)c * N + idn[c]] += 2.f;      // graph_builders.py:67-71
        configs[(long long)c * N + iup[c]] -= 2.f;
        logit[c] = lnew[c];
        if (onsite) onsite[c] = fmaf(2.f, won[idn[c]] - won[iup[c]], onsite[c]);
      }
      if (acc_mask) acc_mask[c] = acc ? 1 : 0;
    }
    s_acc[threadIdx.x] = acc ? 1 : 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int n = 0;
#pragma unroll
    for (int i = 0; i < WIDE_ACC_CHAINS; ++i) n += s_acc[i];
    if (n) atomicAdd(accepted, (unsigned long long)n);
  }
  const int q = Hp >> 2;
  for (int i = threadIdx.x; i < WIDE_ACC_CHAINS * q; i += 256) {
    const int s = i / q, c4 = (i - s * q) * 4;
    if (s_acc[s]) {
      const long long o = (long long)(c0 + s) * Hp + c4;
      *(f32x4*)(z1 + o) = *(const f32x4*)(zc + o);
    }
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

hipError_t launch_wide_propose(hipStream_t s, const float* configs, int B, int N, uint32_t seed_lo,
                               uint32_t seed_hi, int chain_offset, unsigned long long step, const int* inj_up,
                               const int* inj_dn, const float* inj_u, int* iup, int* idn, float* u) {
  hipLaunchKernelGGL(k_wide_propose, dim3((B + 3) / 4), dim3(256), 0, s, configs, B, N, seed_lo, seed_hi,
                     chain_offset, step, inj_up, inj_dn, inj_u, iup, idn, u);
  return hipGetLastError();
}

hipError_t launch_wide_build(hipStream_t s, const float* z1, const float* w1p, const int* iup, const int* idn,
                             int B, int Hp, int act, float* zc, float* a0) {
  hipLaunchKernelGGL(k_wide_build, dim3(blocks_for((long long)B * (Hp / 4))), dim3(256), 0, s, z1, w1p, iup, idn,
                     B, Hp, act, zc, a0);
  return hipGetLastError();
}

hipError_t launch_wide_accept(hipStream_t s, float* configs, float* z1, const float* zc, float* logit,
                              const float* lnew, const int* iup, const int* idn, const float* u, int B, int N,
                              int Hp, int oact, unsigned long long* accepted, unsigned char* acc_mask,
                              float* onsite, const float* won) {
  hipLaunchKernelGGL(k_wide_accept, dim3((B + WIDE_ACC_CHAINS - 1) / WIDE_ACC_CHAINS), dim3(256), 0, s, configs, z1, zc, logit, lnew, iup, idn,
                     u, B, N, Hp, oact, accepted, acc_mask, onsite, won);
  return hipGetLastError();
}

hipError_t launch_wide_delta_last(hipStream_t s, const float* a_last, const float* wout, const float* oscale,
                                  int B, int H, int Hp, int act, float* delta, const float* dact) {
  hipLaunchKernelGGL(k_wide_delta_last, dim3(blocks_for((long long)B * Hp)), dim3(256), 0, s, a_last, wout,
                     oscale, B, H, Hp, act, delta, dact);
  return hipGetLastError();
}
