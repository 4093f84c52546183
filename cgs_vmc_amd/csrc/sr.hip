// Stochastic reconfiguration (named by the north star, ABSENT from the
// reference: training.py only has the plain energy gradient + Adam).  Extension, no reference
// oracle: parity is against the fp64 explicit-S restatement in oracle/vmc_oracle.py.
//
//   S = <O O^T> - <O><O>^T,  f = <E O> - <E><O>,  (S + lambda I) x = f,  theta -= lr x
//
// with O_k(b) = d logit_b / d theta_k over every sample of the epoch.  S is never formed
// (P^2 = 25 G entries at config 3): conjugate gradients with the matrix-free product
//   S v = (1/n) sum_b (O_b . v) O_b - <O> (1/n) sum_b (O_b . v)
// where O_b . v is a forward-mode (tangent) pass through the stored activations and the weighted
// sum is the same [a | 1]^T [t (.) delta] GEMM batch as the energy gradient (grad.hip).  This
// file holds the small kernels around those GEMMs; all reductions are fixed-order.
#include "common.hpp"

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

__device__ __forceinline__ double block_sum_d(double v, double* s) {
  s[threadIdx.x] = v;
  __syncthreads();
  for (int d = blockDim.x >> 1; d >= 1; d >>= 1) {
    if ((int)threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d];
    __syncthreads();
  }
  const double r = s[0];
  __syncthreads();
  return r;
}

// tangent of the logit: t_b = adot_L[b] . w_out + a_L[b] . v_out + v_bout    (one wave per row)
__global__ __launch_bounds__(256) void k_jvp_out(const float* __restrict__ tang,
                                                 const float* __restrict__ act,
                                                 const float* __restrict__ wout,
                                                 const float* __restrict__ vout,
                                                 const float* __restrict__ vbout, int B, int H,
                                                 int Hp, float* __restrict__ t) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float s = 0.f;
  for (int h = lane; h < H; h += 64)
    s += tang[(long long)b * Hp + h] * wout[h] + act[(long long)b * Hp + h] * vout[h];
  s = wave_sum_f(s);
  if (lane == 0) t[b] = s + vbout[0];
}

hipError_t launch_jvp_out(hipStream_t s, const float* tang, const float* act, const float* wout,
                          const float* vout, const float* vbout, int B, int H, int Hp, float* t) {
  hipLaunchKernelGGL(k_jvp_out, dim3((B + 3) / 4), dim3(256), 0, s, tang, act, wout, vout, vbout,
                     B, H, Hp, t);
  return hipGetLastError();
}

// RBM: t_b = zdot_last[b] . tanh(z_last[b]) + x_b . v_on + v_bon   (act holds tanh(z_last))
__global__ __launch_bounds__(256) void k_jvp_out_rbm(const float* __restrict__ tang,
                                                     const float* __restrict__ act,
                                                     const float* __restrict__ cfg,
                                                     const float* __restrict__ von,
                                                     const float* __restrict__ vbon, int B, int H,
                                                     int Hp, int N, float* __restrict__ t) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float s = 0.f;
  for (int h = lane; h < H; h += 64) s = fmaf(tang[(long long)b * Hp + h], act[(long long)b * Hp + h], s);
  for (int n = lane; n < N; n += 64) s = fmaf(cfg[(long long)b * N + n], von[n], s);
  s = wave_sum_f(s);
  if (lane == 0) t[b] = s + vbon[0];
}

hipError_t launch_jvp_out_rbm(hipStream_t s, const float* tang, const float* act,
                              const float* cfg, const float* von, const float* vbon, int B, int H,
                              int Hp, int N, float* t) {
  hipLaunchKernelGGL(k_jvp_out_rbm, dim3((B + 3) / 4), dim3(256), 0, s, tang, act, cfg, von, vbon,
                     B, H, Hp, N, t);
  return hipGetLastError();
}

// dst[0] += sum_b t[b]  (single block, fixed order)
__global__ __launch_bounds__(1024) void k_sum_into(const float* __restrict__ t, int B,
                                                   float* __restrict__ dst) {
  __shared__ double s[1024];
  double a = 0.0;
  for (int i = threadIdx.x; i < B; i += 1024) a += (double)t[i];
  const double r = block_sum_d(a, s);
  if (threadIdx.x == 0) dst[0] += (float)r;
}

hipError_t launch_sum_into(hipStream_t s, const float* t, int B, float* dst) {
  hipLaunchKernelGGL(k_sum_into, dim3(1), dim3(1024), 0, s, t, B, dst);
  return hipGetLastError();
}

// dst[0] = sum of partial[0..n)  (single block, fixed order)
__global__ __launch_bounds__(256) void k_fold(const double* __restrict__ partial, int n,
                                              double* __restrict__ dst, double* __restrict__ dst2) {
  __shared__ double s[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) a += partial[i];
  const double r = block_sum_d(a, s);
  if (threadIdx.x == 0) { dst[0] = r; if (dst2) dst2[0] = r; }
}

#define SR_GRID(P) (int)min(((long long)(P) + 255) / 256, (long long)256)

// right-hand side from the accumulators [g1 | g2 | e_total e_count ...]:
//   f = g2 / n - (e_total / n) g1 / n ;  x = 0, r = p = f ; partial = sum f^2
__global__ __launch_bounds__(256) void k_sr_rhs(const float* __restrict__ acc, int P,
                                                float* __restrict__ x, float* __restrict__ r,
                                                float* __restrict__ p,
                                                double* __restrict__ partial) {
  __shared__ double s[256];
  const float* sc = acc + 2LL * P;
  const float n = sc[1];
  const float mean_e = sc[0] / n;
  double a = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) {
    const float f = acc[P + i] / n - mean_e * (acc[i] / n);
    x[i] = 0.f; r[i] = f; p[i] = f;
    a += (double)f * (double)f;
  }
  const double t = block_sum_d(a, s);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// q = u / n - <O> (u[P] / n) + lambda p ; partial = sum p q
__global__ __launch_bounds__(256) void k_sr_q(const float* __restrict__ u,
                                              const float* __restrict__ acc, int P,
                                              const float* __restrict__ p, float lambda,
                                              float* __restrict__ q,
                                              double* __restrict__ partial) {
  __shared__ double s[256];
  const float n = acc[2LL * P + 1];
  const float tbar = u[P] / n;
  double a = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) {
    const float qi = u[i] / n - (acc[i] / n) * tbar + lambda * p[i];
    q[i] = qi;
    a += (double)p[i] * (double)qi;
  }
  const double t = block_sum_d(a, s);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// alpha = rr / pq ; x += alpha p ; r -= alpha q ; partial = sum r^2
__global__ __launch_bounds__(256) void k_sr_xr(const double* __restrict__ sc, int cur, int P,
                                               const float* __restrict__ p,
                                               const float* __restrict__ q,
                                               float* __restrict__ x, float* __restrict__ r,
                                               double* __restrict__ partial) {
  __shared__ double s[256];
  const double pq = sc[2];
  const float alpha = pq > 0.0 ? (float)(sc[cur] / pq) : 0.f;
  double a = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) {
    x[i] += alpha * p[i];
    const float ri = r[i] - alpha * q[i];
    r[i] = ri;
    a += (double)ri * (double)ri;
  }
  const double t = block_sum_d(a, s);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// beta = rr_new / rr ; p = r + beta p
__global__ __launch_bounds__(256) void k_sr_p(const double* __restrict__ sc, int cur, int P,
                                              const float* __restrict__ r,
                                              float* __restrict__ p) {
  const double rr = sc[cur];
  const float beta = rr > 0.0 ? (float)(sc[cur ^ 1] / rr) : 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256)
    p[i] = r[i] + beta * p[i];
}

__global__ __launch_bounds__(256) void k_sr_apply(float* __restrict__ theta,
                                                  const float* __restrict__ x, float lr, int P) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256)
    theta[i] -= lr * x[i];
}

// sc (double[4]) = [rr ping, rr pong, p.q, rr of the first residual]
hipError_t launch_sr_rhs(hipStream_t s, const float* acc, int P, float* x, float* r, float* p,
                         double* partial, double* sc) {
  const int g = SR_GRID(P);
  hipLaunchKernelGGL(k_sr_rhs, dim3(g), dim3(256), 0, s, acc, P, x, r, p, partial);
  hipLaunchKernelGGL(k_fold, dim3(1), dim3(256), 0, s, partial, g, sc + 0, sc + 3);
  return hipGetLastError();
}

hipError_t launch_sr_q(hipStream_t s, const float* u, const float* acc, int P, const float* p,
                       float lambda, float* q, double* partial, double* sc) {
  const int g = SR_GRID(P);
  hipLaunchKernelGGL(k_sr_q, dim3(g), dim3(256), 0, s, u, acc, P, p, lambda, q, partial);
  hipLaunchKernelGGL(k_fold, dim3(1), dim3(256), 0, s, partial, g, sc + 2, (double*)nullptr);
  return hipGetLastError();
}

hipError_t launch_sr_step(hipStream_t s, double* sc, int cur, int P, float* p, const float* q,
                          float* x, float* r, double* partial) {
  const int g = SR_GRID(P);
  hipLaunchKernelGGL(k_sr_xr, dim3(g), dim3(256), 0, s, sc, cur, P, p, q, x, r, partial);
  hipLaunchKernelGGL(k_fold, dim3(1), dim3(256), 0, s, partial, g, sc + (cur ^ 1),
                     (double*)nullptr);
  hipLaunchKernelGGL(k_sr_p, dim3(g), dim3(256), 0, s, sc, cur, P, r, p);
  return hipGetLastError();
}

// out[0] = sum_b t[b] (single block, fixed order, double partials): u[P] of the convolutional matvec
__global__ __launch_bounds__(1024) void k_sr_tsum(const float* __restrict__ t, int n, float* __restrict__ out) {
  __shared__ double s[1024];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) a += (double)t[i];
  s[threadIdx.x] = a;
  __syncthreads();
  for (int d = 512; d >= 1; d >>= 1) {
    if ((int)threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)s[0];
}

hipError_t launch_sr_tsum(hipStream_t s, const float* t, int n, float* out) {
  hipLaunchKernelGGL(k_sr_tsum, dim3(1), dim3(1024), 0, s, t, n, out);
  return hipGetLastError();
}

hipError_t launch_sr_apply(hipStream_t s, float* theta, const float* x, float lr, int P) {
  hipLaunchKernelGGL(k_sr_apply, dim3(SR_GRID(P)), dim3(256), 0, s, theta, x, lr, P);
  return hipGetLastError();
}
