// Gradient-accumulator path: the two tf.gradients calls of training.py:545-547 / 674-679
// (sum_b O_k(b) and sum_b w_b O_k(b), O_k = d logit / d theta_k) as an explicit
// forward / back-prop / weight-gradient GEMM chain on fp32 MFMA, plus the accumulator,
// ratio and Adam element-wise kernels.  All reductions are fixed-order (no float atomics).
#include "common.hpp"

// ----------------------------------------------------------------------------------- GEMM
// C[M,N] = A[M,K] B[K,N] with arbitrary element strides, 64x64x16 block tile, 4 waves in a
// 2x2 grid, one 32x32 v_mfma_f32_32x32x2_f32 accumulator per wave.  Optional split-K over
// blockIdx.z into a workspace that k_gemm_reduce folds in z order.
#define GT 64
#define GK 16
#define GLD 65

__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, int m, int n, float v) {
  float* c = g.C + (long long)m * g.ldc + n;
  switch (g.epilogue) {
    case 1: *c = fmaxf(v + g.bias[n], 0.f); break;
    case 2: *c = g.mask[(long long)m * g.ldmask + n] > 0.f ? v : 0.f; break;
    case 3: *c += v; break;
    default: *c = v; break;
  }
}

__global__ __launch_bounds__(256) void k_gemm(GemmArgs g) {
  __shared__ float As[GK][GLD];
  __shared__ float Bs[GK][GLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  int kc = (g.K + g.splitk - 1) / g.splitk;
  kc = (kc + GK - 1) / GK * GK;
  const int kbeg = blockIdx.z * kc;
  const int kend = min(g.K, kbeg + kc);
  const bool a_kfast = (g.sak == 1), b_kfast = (g.sbk == 1);

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (int k0 = kbeg; k0 < kend; k0 += GK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;
      int m, k;
      if (a_kfast) { k = idx % GK; m = idx / GK; } else { m = idx % GT; k = idx / GT; }
      const int gm = m0 + m, gk = k0 + k;
      As[k][m] = (gm < g.M && gk < kend) ? g.A[(long long)gm * g.sam + (long long)gk * g.sak] : 0.f;
      int n, kb;
      if (b_kfast) { kb = idx % GK; n = idx / GK; } else { n = idx % GT; kb = idx / GT; }
      const int gn = n0 + n, gkb = k0 + kb;
      float bv = 0.f;
      if (gn < g.N && gkb < kend) {
        bv = g.B[(long long)gkb * g.sbk + (long long)gn * g.sbn];
        if (g.kscale) bv *= g.kscale[gkb];
      }
      Bs[kb][n] = bv;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GK; kk += 2) {
      const float av = As[kk + (lane >> 5)][wm * 32 + (lane & 31)];
      const float bv = Bs[kk + (lane >> 5)][wn * 32 + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    __syncthreads();
  }

  const int n = n0 + wn * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (m < g.M && n < g.N) {
      if (g.splitk > 1) {
        g.workspace[((long long)blockIdx.z * g.M + m) * g.N + n] = acc[r];
      } else {
        gemm_epilogue(g, m, n, acc[r]);
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_gemm_reduce(GemmArgs g) {
  const long long total = (long long)g.M * g.N;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (long long)gridDim.x * 256) {
    float v = 0.f;
    for (int z = 0; z < g.splitk; ++z) v += g.workspace[(long long)z * total + i];
    gemm_epilogue(g, (int)(i / g.N), (int)(i % g.N), v);
  }
}

hipError_t launch_gemm(hipStream_t s, const GemmArgs& g) {
  if (g.M <= 0 || g.N <= 0) return hipSuccess;
  const dim3 grid((g.N + GT - 1) / GT, (g.M + GT - 1) / GT, g.splitk);
  hipLaunchKernelGGL(k_gemm, grid, dim3(256), 0, s, g);
  if (g.splitk > 1) {
    const long long total = (long long)g.M * g.N;
    const int blocks = (int)min((total + 255) / 256, (long long)2048);
    hipLaunchKernelGGL(k_gemm_reduce, dim3(blocks), dim3(256), 0, s, g);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------- element-wise
__global__ void k_relu_copy(const float* __restrict__ z, float* __restrict__ a, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    a[i] = fmaxf(z[i], 0.f);
}

hipError_t launch_relu_copy(hipStream_t s, const float* z, float* a, long long n) {
  const int blocks = (int)min((n + 255) / 256, (long long)4096);
  hipLaunchKernelGGL(k_relu_copy, dim3(blocks), dim3(256), 0, s, z, a, n);
  return hipGetLastError();
}

__global__ void k_fill(float* __restrict__ x, float v, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    x[i] = v;
}

hipError_t launch_fill(hipStream_t s, float* x, float v, long long n) {
  if (n <= 0) return hipSuccess;
  const int blocks = (int)min((n + 255) / 256, (long long)4096);
  hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, s, x, v, n);
  return hipGetLastError();
}

// delta_L[b][i] = d logit_b / d z_L[b][i] = w_out[i] * relu'(z_L) (a_L > 0 <=> z_L > 0)
__global__ void k_delta_out(const float* __restrict__ woutp, const float* __restrict__ aL,
                            float* __restrict__ delta, int B, int Hp) {
  const long long n = (long long)B * Hp;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    delta[i] = aL[i] > 0.f ? woutp[i % Hp] : 0.f;
}

hipError_t launch_delta_out(hipStream_t s, const float* woutp, const float* aL, float* delta,
                            int B, int Hp) {
  const long long n = (long long)B * Hp;
  const int blocks = (int)min((n + 255) / 256, (long long)4096);
  hipLaunchKernelGGL(k_delta_out, dim3(blocks), dim3(256), 0, s, woutp, aL, delta, B, Hp);
  return hipGetLastError();
}

// weighted column sums: out[c] += sum_b w[b] X[b*ld + c]; two fixed-order stages
#define WCS_CHUNKS 32
__global__ __launch_bounds__(256) void k_wcolsum1(const float* __restrict__ X, long long ld,
                                                  const float* __restrict__ w, int B, int ncols,
                                                  float* __restrict__ ws) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int per = (B + WCS_CHUNKS - 1) / WCS_CHUNKS;
  const int b0 = blockIdx.y * per, b1 = min(B, b0 + per);
  if (c >= ncols) return;
  float acc = 0.f;
  for (int b = b0; b < b1; ++b) acc = fmaf(w ? w[b] : 1.f, X[(long long)b * ld + c], acc);
  ws[(long long)blockIdx.y * ncols + c] = acc;
}

__global__ __launch_bounds__(256) void k_wcolsum2(const float* __restrict__ ws, int ncols,
                                                  float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncols) return;
  float acc = 0.f;
  for (int s = 0; s < WCS_CHUNKS; ++s) acc += ws[(long long)s * ncols + c];
  out[c] += acc;
}

hipError_t launch_wcolsum(hipStream_t s, const float* X, long long ld, const float* w, int B,
                          int ncols, float* out, float* workspace) {
  const dim3 grid((ncols + 255) / 256, WCS_CHUNKS);
  hipLaunchKernelGGL(k_wcolsum1, grid, dim3(256), 0, s, X, ld, w, B, ncols, workspace);
  hipLaunchKernelGGL(k_wcolsum2, dim3((ncols + 255) / 256), dim3(256), 0, s, workspace, ncols,
                     out);
  return hipGetLastError();
}

// tf.metrics.mean updates (training.py:555, 689-690) + mean_tensor count (550-553):
// scalars = [e_total, e_count, r_total, r_count, g_count]
__global__ __launch_bounds__(1024) void k_scalar_accum(const float* __restrict__ eloc,
                                                       const float* __restrict__ ratio, int B,
                                                       float* __restrict__ sc, int mode) {
  __shared__ double se[1024];
  __shared__ double sr[1024];
  double e = 0.0, r = 0.0;
  for (int i = threadIdx.x; i < B; i += 1024) {
    e += (double)eloc[i];
    if (ratio) r += (double)ratio[i];
  }
  se[threadIdx.x] = e; sr[threadIdx.x] = r;
  __syncthreads();
  for (int d = 512; d >= 1; d >>= 1) {
    if (threadIdx.x < d) { se[threadIdx.x] += se[threadIdx.x + d]; sr[threadIdx.x] += sr[threadIdx.x + d]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    sc[0] += (float)se[0];
    sc[1] += (float)B;
    if (mode == 1) { sc[2] += (float)sr[0]; sc[3] += (float)B; }
    sc[4] += 1.f;
  }
}

hipError_t launch_scalar_accum(hipStream_t s, const float* eloc, const float* ratio, int B,
                               float* acc_scalars, int mode) {
  hipLaunchKernelGGL(k_scalar_accum, dim3(1), dim3(1024), 0, s, eloc, ratio, B, acc_scalars,
                     mode);
  return hipGetLastError();
}

// ratio_b = (psi_w - beta H psi_w)/psi  (training.py:665-672)
//         = exp(logit_w - logit_psi + shift_psi - shift_w) * (1 - beta * E_loc^w)
__global__ void k_itswo_ratio(const float* __restrict__ lp, const float* __restrict__ lw,
                              const float* __restrict__ ew, float log_factor, float beta, int B,
                              float* __restrict__ ratio) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) ratio[i] = expf(lw[i] - lp[i] + log_factor) * (1.f - beta * ew[i]);
}

hipError_t launch_itswo_ratio(hipStream_t s, const float* logit_psi, const float* logit_omega,
                              const float* eloc_omega, float log_factor, float beta, int B,
                              float* ratio) {
  hipLaunchKernelGGL(k_itswo_ratio, dim3((B + 255) / 256), dim3(256), 0, s, logit_psi,
                     logit_omega, eloc_omega, log_factor, beta, B, ratio);
  return hipGetLastError();
}

// gradient formula (training.py:560-564 / 697-699) + TF1 Adam (training.py:84-91)
//   acc = [g1 (P) | g2 (P) | e_total e_count r_total r_count g_count ...]
__global__ void k_adam(float* __restrict__ theta, float* __restrict__ m, float* __restrict__ v,
                       const float* __restrict__ acc, int P, int mode, float lr_t, float b1,
                       float b2, float eps, float* __restrict__ grad_out, int apply) {
  const float* sc = acc + 2LL * P;
  const float gc = sc[4];
  const float mean_e = sc[0] / sc[1];
  const float mean_r = sc[2] / sc[3];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
    const float g1 = acc[i] / gc, g2 = acc[P + i] / gc;
    const float g = mode == 0 ? g2 - mean_e * g1 : g1 - g2 / mean_r;
    if (grad_out) grad_out[i] = g;
    if (apply) {
      const float mi = m[i] + (g - m[i]) * (1.f - b1);
      const float vi = v[i] + (g * g - v[i]) * (1.f - b2);
      m[i] = mi; v[i] = vi;
      theta[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
  }
}

hipError_t launch_adam(hipStream_t s, float* theta, float* m, float* v, const float* acc, int P,
                       int mode, float lr_t, float b1, float b2, float eps, float* grad_out) {
  const int blocks = min((P + 255) / 256, 2048);
  hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, s, theta, m, v, acc, P, mode, lr_t, b1,
                     b2, eps, grad_out, theta != nullptr ? 1 : 0);
  return hipGetLastError();
}
