// Connected-configuration list and reductions of HeisenbergHamiltonian.build /
// local_value (operators.py:137-169, 227-259) for gfx950.
//
// The reference evaluates psi(swap_ij R) for EVERY bond and multiplies by the mask
// [s_i s_j < 0] afterwards (operators.py:166-168).  Here the masked-out rows are never
// generated: k_bond_count / k_bond_fill build a compact, chain-ordered list of the
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
  // four 64-bond groups per pass, their loads issued together (bonds, then spins): two dependent round
  // trips per 256 bonds instead of eight (the loop was the kernel's whole time); same per-lane order of the
  // fused multiply-adds as one group per pass
  for (int k0 = 0; k0 < n_bonds; k0 += 256) {
    int2 ab[4]; float q[4], xi[4], xj[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + 64 * u + lane, kk = k < n_bonds ? k : 0;
      ab[u] = bonds[kk]; q[u] = quarter_jz[kk];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { xi[u] = x[ab[u].x]; xj[u] = x[ab[u].y]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool in = k0 + 64 * u + lane < n_bonds;
      const float sz = xi[u] * xj[u];
      if (in) d = fmaf(q[u], sz, d);
      n += __popcll(__ballot(in && sz < 0.f));
    }
  }
  d = wave_sum(d);
  if (lane == 0) { cnt[c] = n; diag[c] = d; }
}

// 16 chains per workgroup (one wave each): the workgroup first derives the exclusive prefix of the
// per-chain counts up to its own chains itself -- at most B integers from L2, summed by its 1024
// threads in a fixed order -- instead of waiting for a separate single-block scan launch; it writes
// off[c] for its chains (k_eloc_reduce and the row kernels read them; the last workgroup also
// off[B] = total), then rowinfo[off[c] + p] = {c, +-(bond+1)}, sign = sign of s_i, bonds in
// ascending order.
__global__ __launch_bounds__(1024) void k_bond_fill(const float* __restrict__ configs,
                                                    const int2* __restrict__ bonds, int B, int N,
                                                    int n_bonds, const int* __restrict__ cnt,
                                                    int* __restrict__ off,
                                                    int2* __restrict__ rowinfo) {
  __shared__ int s_wave[16];
  __shared__ int s_cnt[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int c0 = blockIdx.x * 16;
  // the first 256 bonds of this wave's chain travel while the prefix is summed (they do not depend on it)
  const int cw = c0 + wave < B ? c0 + wave : B - 1;
  const float* x = configs + (long long)cw * N;
  int2 ab0[4]; float xi0[4], xj0[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { const int k = 64 * u + lane; ab0[u] = bonds[k < n_bonds ? k : 0]; }
#pragma unroll
  for (int u = 0; u < 4; ++u) { xi0[u] = x[ab0[u].x]; xj0[u] = x[ab0[u].y]; }
  // sum of cnt[0 .. c0): thread t takes the elements t, t + 1024, ...
  int part = 0;
  for (int i = t; i < c0; i += 1024) part += cnt[i];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
  if (lane == 0) s_wave[wave] = part;
  if (t < 16) s_cnt[t] = c0 + t < B ? cnt[c0 + t] : 0;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) base += s_wave[w];
  for (int w = 0; w < wave; ++w) base += s_cnt[w];      // exclusive prefix inside the workgroup
  const int c = c0 + wave;
  if (c >= B) return;
  if (lane == 0) {
    off[c] = base;
    if (c == B - 1) off[B] = base + s_cnt[wave];
  }
  for (int k0 = 0; k0 < n_bonds; k0 += 64) {
    const int k = k0 + lane;
    bool anti = false;
    float si = 0.f;
    if (k0 < 256) {                               // (uniform) the prefetched groups
      const int u = k0 >> 6;
      si = u == 0 ? xi0[0] : (u == 1 ? xi0[1] : (u == 2 ? xi0[2] : xi0[3]));
      const float sj = u == 0 ? xj0[0] : (u == 1 ? xj0[1] : (u == 2 ? xj0[2] : xj0[3]));
      anti = k < n_bonds && si * sj < 0.f;
    } else if (k < n_bonds) {
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
                            int* off, float* diag, int2* rowinfo, bool counted) {
  if (!counted)      // (the sampler's last launch has left the census of these chains: sweep16.hpp)
    hipLaunchKernelGGL(k_bond_count, dim3((B + 3) / 4), dim3(256), 0, s, configs, bonds, quarter_jz, B, N, n_bonds,
                       cnt, diag);
  hipLaunchKernelGGL(k_bond_fill, dim3((B + 15) / 16), dim3(1024), 0, s, configs, bonds, B, N, n_bonds, cnt, off,
                     rowinfo);
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

__global__ void k_div_f64(double* __restrict__ x, int n, double d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = x[i] / d;       // IEEE division: the same bits as the host's sum / B
}

hipError_t launch_div_f64(hipStream_t s, double* x, int n, double d) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_div_f64, dim3((n + 255) / 256), dim3(256), 0, s, x, n, d);
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
