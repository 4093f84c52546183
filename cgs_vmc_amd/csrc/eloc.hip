// Connected-configuration list and reductions of HeisenbergHamiltonian.build /
// local_value (operators.py:137-169, 227-259) for gfx950.
//
// The reference evaluates psi(swap_ij R) for EVERY bond and multiplies by the mask
// [s_i s_j < 0] afterwards (operators.py:166-168).  Here the masked-out rows are never
// generated: k_bond_count / k_scan / k_bond_fill build a compact, chain-ordered list of the
// antiparallel bonds, k_tail16 (mlp.hip) evaluates exactly those rows, and k_eloc_reduce sums
// each chain's segment in a fixed order (deterministic, no float atomics).
#include "common.hpp"

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// one wave per chain: number of antiparallel bonds and the diagonal term
// diag = sum_bonds 0.25 jz s_i s_j  (operators.py:165, 169, 247)
__global__ __launch_bounds__(256) void k_bond_count(const float* __restrict__ configs,
                                                    const int2* __restrict__ bonds,
                                                    const float* __restrict__ quarter_jz, int B,
                                                    int N, int n_bonds, int* __restrict__ cnt,
                                                    float* __restrict__ diag) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= B) return;
  const float* x = configs + (long long)c * N;
  float d = 0.f;
  int n = 0;
  for (int k0 = 0; k0 < n_bonds; k0 += 64) {
    const int k = k0 + lane;
    bool anti = false;
    if (k < n_bonds) {
      const int2 ab = bonds[k];
      const float sz = x[ab.x] * x[ab.y];
      d = fmaf(quarter_jz[k], sz, d);
      anti = sz < 0.f;
    }
    n += __popcll(__ballot(anti));
  }
  d = wave_sum(d);
  if (lane == 0) { cnt[c] = n; diag[c] = d; }
}

// single block: off[c] = sum_{c' < c} cnt[c'], off[B] = total.  Thread t owns the contiguous
// slice [t*per, (t+1)*per): serial sum, wave-level inclusive scan of the 64 slice totals by
// shuffles, the 16 wave totals through LDS, then the slice is written out (two barriers in all).
__global__ __launch_bounds__(1024) void k_scan(const int* __restrict__ cnt, int B,
                                               int* __restrict__ off) {
  __shared__ int s_wave[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int per = (B + 1023) / 1024;
  const int beg = min(t * per, B), end = min(beg + per, B);
  int total = 0;
  for (int i = beg; i < end; ++i) total += cnt[i];
  int incl = total;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += s_wave[w];
  int run = base + incl - total;          // exclusive prefix of this thread's slice
  for (int i = beg; i < end; ++i) { off[i] = run; run += cnt[i]; }
  if (t == 1023) off[B] = base + incl;
}

// one wave per chain: rowinfo[off[c] + p] = {c, +-(bond+1)}, sign = sign of s_i, bonds in
// ascending order
__global__ __launch_bounds__(256) void k_bond_fill(const float* __restrict__ configs,
                                                   const int2* __restrict__ bonds, int B, int N,
                                                   int n_bonds, const int* __restrict__ off,
                                                   int2* __restrict__ rowinfo) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= B) return;
  const float* x = configs + (long long)c * N;
  int base = off[c];
  for (int k0 = 0; k0 < n_bonds; k0 += 64) {
    const int k = k0 + lane;
    bool anti = false;
    float si = 0.f;
    if (k < n_bonds) {
      const int2 ab = bonds[k];
      si = x[ab.x];
      anti = si * x[ab.y] < 0.f;
    }
    const unsigned long long m = __ballot(anti);
    if (anti) {
      const int p = __popcll(m & ((1ull << lane) - 1ull));
      rowinfo[base + p] = make_int2(c, si > 0.f ? (k + 1) : -(k + 1));
    }
    base += __popcll(m);
  }
}

hipError_t launch_bond_list(hipStream_t s, const float* configs, const int2* bonds,
                            const float* quarter_jz, int B, int N, int n_bonds, int* cnt,
                            int* off, float* diag, int2* rowinfo) {
  const dim3 grid((B + 3) / 4), block(256);
  hipLaunchKernelGGL(k_bond_count, grid, block, 0, s, configs, bonds, quarter_jz, B, N, n_bonds,
                     cnt, diag);
  hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, cnt, B, off);
  hipLaunchKernelGGL(k_bond_fill, grid, block, 0, s, configs, bonds, B, N, n_bonds, off, rowinfo);
  return hipGetLastError();
}

// one wave per chain: offdiag[c] = sum of the chain's rows (0.5 jx psi'/psi each, written by
// k_tail16), eloc[c] = diag[c] + offdiag[c]   (operators.py:259)
__global__ __launch_bounds__(256) void k_eloc_reduce(const int* __restrict__ off,
                                                     const float* __restrict__ diag,
                                                     const float* __restrict__ val, int B,
                                                     float* __restrict__ offdiag,
                                                     float* __restrict__ eloc) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= B) return;
  const int r0 = off[c], r1 = off[c + 1];
  float s = 0.f;
  for (int r = r0 + lane; r < r1; r += 64) s += val[r];
  s = wave_sum(s);
  if (lane == 0) {
    if (offdiag) offdiag[c] = s;
    eloc[c] = diag[c] + s;
  }
}

hipError_t launch_eloc_reduce(hipStream_t s, const int* off, const float* diag, const float* val,
                              int B, float* offdiag, float* eloc) {
  hipLaunchKernelGGL(k_eloc_reduce, dim3((B + 3) / 4), dim3(256), 0, s, off, diag, val, B,
                     offdiag, eloc);
  return hipGetLastError();
}

// deterministic single-block reductions
__global__ __launch_bounds__(1024) void k_sum(const float* __restrict__ x, int n,
                                              double* __restrict__ out) {
  __shared__ double s[1024];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) acc += (double)x[i];
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int d = 512; d >= 1; d >>= 1) {
    if (threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = s[0];
}

__global__ __launch_bounds__(1024) void k_max(const float* __restrict__ x, int n,
                                              float* __restrict__ out) {
  __shared__ float s[1024];
  float acc = -INFINITY;
  for (int i = threadIdx.x; i < n; i += 1024) acc = fmaxf(acc, x[i]);
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int d = 512; d >= 1; d >>= 1) {
    if (threadIdx.x < d) s[threadIdx.x] = fmaxf(s[threadIdx.x], s[threadIdx.x + d]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = s[0];
}

hipError_t launch_sum(hipStream_t s, const float* x, int n, double* out_sum) {
  hipLaunchKernelGGL(k_sum, dim3(1), dim3(1024), 0, s, x, n, out_sum);
  return hipGetLastError();
}

hipError_t launch_max(hipStream_t s, const float* x, int n, float* out_max) {
  hipLaunchKernelGGL(k_max, dim3(1), dim3(1024), 0, s, x, n, out_max);
  return hipGetLastError();
}

// vmc_set_configs: every entry must be exactly +1 or -1 (the sampler's argmax/argmin proposal
// and the bond mask assume it); flag != 0 on violation
__global__ void k_check_pm1(const float* __restrict__ x, long long n, int* __restrict__ flag) {
  int bad = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float v = x[i];
    bad |= (v != 1.f && v != -1.f) ? 1 : 0;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

hipError_t launch_check_pm1(hipStream_t s, const float* x, long long n, int* flag) {
  hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), s);
  if (e != hipSuccess) return e;
  const long long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_check_pm1, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, s, x, n, flag);
  return hipGetLastError();
}
