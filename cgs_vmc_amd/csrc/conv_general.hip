// Convolutional ansatz types beyond the limits of the fused kernels (conv_kernels.hpp keeps a sample's feature maps
// in LDS: kernel_size <= 9, num_conv_filters <= 64, two maps within 160 KiB): the general path.  The reference takes
// any value (wavefunctions.py:534-579, utils.py:107-111).  Feature maps live in HBM, channel-last
// [row][site][Fp] (Fp = filters padded to 4); one Conv2dPeriodic / Conv1dPeriodic (layers.py:24-160) is
//   * a gather  A[(row, site)][(tap, c)] = in[row][(site + tap - lo) mod lattice][c]  -- the periodic padding of
//     layers.py:118-148 / 51-74 is the index arithmetic -- either written out as an im2col matrix (k_cgen_im2col) or, for
//     64 .. filters in multiples of 32, performed by the ring GEMM's A-operand DMA itself (GemmArgs.conv_a, grad.hip);
//     the stored maps hold ACTIVATIONS (f(z_l) behind a convolution, selu(u) inside a residual block, layers.py:226), so
//     that nothing has to be applied on the way; only the cosine, whose derivative needs z, keeps pre-activations and
//     has f applied as k_cgen_im2col gathers --
//   * ONE product with the parameter slice as it lies in theta: snt.Conv2D's w[k, k, Cin, F] IS the row-major
//     [k k Cin][F] B matrix (launch_gemm: k_gemm_ring / k_gemm128 / k_gemm by shape), bias and activation in the
//     epilogue, the residual add of ResBlock2d (layers.py:227) as the accumulate-into-C epilogue.
// The first convolution gathers from the spins themselves, with the exchanged pair of a connected configuration
// (operators.py:162-163) or of a proposed move (graph_builders.py:67-71) negated on the fly.  The logit is the sum of
// the last map (wavefunctions.py:569, 577; 760, 773) in double, one workgroup per row.
// Same arithmetic as the fused kernels up to the order of additions; same Philox streams, same accept rule.
#include "conv.hpp"

namespace {

__device__ __forceinline__ float cg_selu(float x) {   // layers.py:226 tf.nn.selu; constants as the fused kernels / the oracle
  const float scale = 1.0507009873554805f, alpha = 1.6732632423543772f;
  return scale * (x > 0.f ? x : alpha * (expf(x) - 1.f));
}
__device__ __forceinline__ float cg_pre(int pre, float x) {
  return pre < 0 ? x : (pre == CGEN_PRE_SELU ? cg_selu(x) : vmc_act_rt(pre, x));
}

// site index of tap t at position n: ((a1 + d1 - lo) mod D1, (a2 + d2 - lo2) mod D2), layers.py:132-141 / 66-72
// inverse: the position whose tap t reads site n, ((a1 - d1 + lo) mod D1, (a2 - d2 + lo2) mod D2) -- the gather of
// the transposed convolution (d / d input from d / d output)
__device__ __forceinline__ int cg_site(const ConvGeom& g, int n, int t, bool inverse = false) {
  const int a1 = n / g.D2, a2 = n - a1 * g.D2;
  const int d1 = t / g.KW, d2 = t - d1 * g.KW;
  int s1 = (inverse ? a1 - d1 + g.lo : a1 + d1 - g.lo) % g.D1; if (s1 < 0) s1 += g.D1;
  int s2 = (inverse ? a2 - d2 + g.lo2 : a2 + d2 - g.lo2) % g.D2; if (s2 < 0) s2 += g.D2;
  return s1 * g.D2 + s2;
}

// convolutions 1 ..: A[m][t F + c] = f(in[row][site(n, t)][c]); VEC = 4: F % 4 == 0 (16-byte pieces), else scalar.
// A workgroup takes `ppb` consecutive positions (>= 512 items of VEC floats); every index is 32-bit arithmetic (a
// 64-bit division per item made the first form of this kernel VALU-bound: 3.2 TB/s of 165 us per layer).
template <int VEC>
__global__ __launch_bounds__(256) void k_cgen_im2col(CgenIm2colArgs a, int ppb) {
  const ConvGeom g = a.g;
  const int T = g.K * g.KW, C = g.F / VEC, items = T * C;
  const int M = a.rows * g.N;                                   // (rows * N < 2^31: the M of the GEMM that follows)
  for (int m0 = blockIdx.x * ppb; m0 < M; m0 += gridDim.x * ppb) {
    const int np = min(ppb, M - m0);
    for (int w = threadIdx.x; w < np * items; w += 256) {
      const int pl = w / items, it = w - pl * items;
      const int t = it / C, cq = it - t * C;
      const int m = m0 + pl;
      const int r = m / g.N, n = m - r * g.N;
      const float* src = a.src + ((long long)r * g.N + cg_site(g, n, t, a.inverse != 0)) * a.Fp + VEC * cq;
      float* dst = a.A + (long long)m * a.lda + t * g.F + VEC * cq;
      if (VEC == 4) {
        f32x4 v = *(const f32x4*)src;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = cg_pre(a.pre_act, v[e]);
        *(f32x4*)dst = v;
      } else {
        *dst = cg_pre(a.pre_act, *src);
      }
    }
  }
}

// first convolution: A[m][t] = s'(site(n, t)) with s' = the row's spins, the exchanged pair negated
__global__ __launch_bounds__(256) void k_cgen_im2col0(CgenIm2colArgs a) {
  const ConvGeom g = a.g;
  const int T = g.K * g.KW;
  const int M = a.rows * g.N;
  const long long total = (long long)M * T;
  for (long long base = (long long)blockIdx.x * 256; base < total; base += (long long)gridDim.x * 256) {
    const long long idx = base + threadIdx.x;
    if (idx >= total) break;
    const int m = (int)(idx / T), t = (int)(idx - (long long)m * T);
    const int r = m / g.N, n = m - r * g.N;
    int chain = (int)a.row0 + r, fa = -1, fb = -1;
    if (a.rowinfo) {
      const int2 ri = a.rowinfo[a.row0 + r];
      chain = ri.x;
      if (ri.y != 0) { const int2 ab = a.bonds[(ri.y > 0 ? ri.y : -ri.y) - 1]; fa = ab.x; fb = ab.y; }
    }
    if (a.iup) { fa = a.iup[a.row0 + r]; fb = a.idn[a.row0 + r]; }
    const int s = cg_site(g, n, t);
    const float x = a.src[(long long)chain * g.N + s];
    a.A[(long long)m * a.lda + t] = (s == fa || s == fb) ? -x : x;
  }
}

// sum of a row's last feature map over sites and channels (wavefunctions.py:569 reduce_sum), in double: one workgroup
// per row, 16-byte loads, a fixed order (thread-strided partial sums, xor tree per wave, the four waves in order)
__global__ __launch_bounds__(256) void k_cgen_rowsum(const float* __restrict__ fm, int rows, int N, int F, int Fp,
                                                     double* __restrict__ out) {
  __shared__ double s_w[4];
  const int r = blockIdx.x;
  const float* p = fm + (long long)r * N * Fp;
  const int q = N * Fp / 4, fq = Fp / 4;
  double s = 0.0;
  for (int i = threadIdx.x; i < q; i += 256) {
    const f32x4 v = *(const f32x4*)(p + 4 * i);
    const int c0 = 4 * (i % fq);
#pragma unroll
    for (int e = 0; e < 4; ++e) s += c0 + e < F ? (double)v[e] : 0.0;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[r] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

// Metropolis test and commit of one mc_step (graph_builders.py:75-88): one thread per chain, one atomic per workgroup
__global__ __launch_bounds__(256) void k_cgen_accept(float* __restrict__ configs, float* __restrict__ logit,
                                                     const float* __restrict__ lnew, const int* __restrict__ iup,
                                                     const int* __restrict__ idn, const float* __restrict__ u, int B,
                                                     int N, int oact, unsigned long long* __restrict__ accepted,
                                                     unsigned char* __restrict__ acc_mask) {
  __shared__ int s_n;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  bool acc = false;
  if (c < B) {
    const float uu = u[c];
    acc = vmc_out_accept(oact, lnew[c], logit[c], uu, 0.5f * __logf(uu));
    if (acc) {
      configs[(long long)c * N + idn[c]] += 2.f;      // graph_builders.py:67-71
      configs[(long long)c * N + iup[c]] -= 2.f;
      logit[c] = lnew[c];
    }
    if (acc_mask) acc_mask[c] = acc ? 1 : 0;
  }
  const unsigned long long b = __ballot(acc);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(&s_n, __popcll(b));
  __syncthreads();
  if (threadIdx.x == 0 && s_n) atomicAdd(accepted, (unsigned long long)s_n);
}

// The end of one mc_step and the start of the next in ONE launch (round 6): the sum of chain c's last map (k_cgen_rowsum's
// arithmetic: thread-strided double sums, xor tree per wave, the four waves in order), the candidate's logit
// (k_wide_out_part with one partial and a zero output bias: (float)sum + 0), the Metropolis test and commit
// (k_cgen_accept), and the NEXT step's proposal from the chain as it then stands (k_wide_propose's arithmetic: one wave,
// strict comparisons in site-block order, the lowest index among equal values across lanes) -- four launches of
// 5 - 10 us each per step where a step's three convolutions take ~35 (36 x 36 sites, 32 chains).  One workgroup per chain.
__global__ __launch_bounds__(256) void k_cgen_step_tail(const float* __restrict__ fm, int N, int F, int Fp,
                                                        float* __restrict__ configs, float* __restrict__ logit, int oact,
                                                        int* __restrict__ iup, int* __restrict__ idn, float* __restrict__ u,
                                                        unsigned long long* __restrict__ accepted, uint32_t seed_lo,
                                                        uint32_t seed_hi, int chain_offset, unsigned long long next_step,
                                                        int do_propose) {
  __shared__ double s_w[4];
  __shared__ int s_flip[3];                       // accepted, the site raised, the site lowered
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const float* p = fm + (long long)c * N * Fp;
  const int q = N * Fp / 4, fq = Fp / 4;
  double s = 0.0;
  for (int i = tid; i < q; i += 256) {
    const f32x4 v = *(const f32x4*)(p + 4 * i);
    const int c0 = 4 * (i % fq);
#pragma unroll
    for (int e = 0; e < 4; ++e) s += c0 + e < F ? (double)v[e] : 0.0;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  if (lane == 0) s_w[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) {
    const double sd = 0.0 + ((s_w[0] + s_w[1]) + (s_w[2] + s_w[3]));
    const float lnew = (float)sd + 0.f;
    const float uu = u[c];
    const bool acc = vmc_out_accept(oact, lnew, logit[c], uu, 0.5f * __logf(uu));
    const int dn = idn[c], up = iup[c];
    if (acc) {
      configs[(long long)c * N + dn] += 2.f;      // graph_builders.py:67-71
      configs[(long long)c * N + up] -= 2.f;
      logit[c] = lnew;
      atomicAdd(accepted, 1ull);
    }
    s_flip[0] = acc ? 1 : 0; s_flip[1] = dn; s_flip[2] = up;
  }
  __syncthreads();
  if (!do_propose || tid >= 64) return;
  // the next proposal (graph_builders.py:59-65) from the committed chain: the spins as loaded, the accepted pair patched in
  const bool acc = s_flip[0] != 0;
  const int dn = s_flip[1], up = s_flip[2];
  const float* x = configs + (long long)c * N;
  const uint2 key = make_uint2(seed_lo, seed_hi);
  const uint32_t gid = (uint32_t)(chain_offset + c);
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
        float xi = x[i];
        // (thread 0's stores above may or may not have reached this load: the value is taken from before the move and the
        // move applied here -- spins are +-1, so x +- 2 is exact either way round)
        if (acc && (i == dn || i == up)) xi = i == dn ? 1.f : -1.f;
        const float v = xi * u32_to_uniform(rr[e]);
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
    iup[c] = idx_hi; idn[c] = idx_lo; u[c] = u32_to_uniform(ra.x);
  }
}

// ---- gradient path
__device__ __forceinline__ float cg_dpre(int pre, float z) {    // f'(z) of the gather's activation
  if (pre < 0) return 1.f;
  if (pre == CGEN_PRE_SELU) {
    const float scale = 1.0507009873554805f, alpha = 1.6732632423543772f;
    return z > 0.f ? scale : scale * alpha * expf(z);
  }
  return vmc_dact_rt(pre, z, vmc_act_rt(pre, z));
}

// d logit / d (last map) = the per-sample output factor (1 for the exp output); padding channels 0
__global__ void k_cgen_fill(float* __restrict__ gm, const float* __restrict__ oscale, long long row0, int rows, int N,
                            int F, int Fp) {
  const long long total = (long long)rows * N * Fp;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / ((long long)N * Fp));
    gm[i] = (int)(i % Fp) < F ? (oscale ? oscale[row0 + r] : 1.f) : 0.f;
  }
}

// out = d (.) f'  (d / d pre-activation from d / d activation; f' off the stored z or off the stored activation); padding channels 0
// from_act: the map holds a = f(z) (cgen_post): f' read off the activation (selu: off t = selu(u))
__device__ __forceinline__ float cg_dpre_from_act(int pre, float a) {
  if (pre < 0) return 1.f;
  if (pre == CGEN_PRE_SELU) return a > 0.f ? 1.0507009873554805f : a + 1.0507009873554805f * 1.6732632423543772f;
  return vmc_dact_rt(pre, a, a);                    // (relu: a > 0 <=> z > 0; the cosine never comes here)
}
__global__ void k_cgen_dact(const float* __restrict__ d, const float* __restrict__ z, int pre, int from_act, long long n,
                            int F, int Fp, float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = (int)(i % Fp) < F ? d[i] * (from_act ? cg_dpre_from_act(pre, z[i]) : cg_dpre(pre, z[i])) : 0.f;
}

// per-position copy of the per-sample weights: the k-scale of the weight-gradient products
__global__ void k_cgen_wpos(const float* __restrict__ w, long long row0, int rows, int N, float* __restrict__ wpos) {
  const long long total = (long long)rows * N;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    wpos[i] = w[row0 + i / N];
}

// transposed weight image of convolution l >= 1: wt[(t F + o) F + c] = w[(t F + c) F + o]
__global__ void k_cgen_pack_t(const float* __restrict__ w, int T, int F, float* __restrict__ wt) {
  const long long total = (long long)T * F * F;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % F);
    const long long to = i / F;
    const int o = (int)(to % F), t = (int)(to / F);
    wt[i] = w[((long long)t * F + c) * F + o];
  }
}

// ---- stochastic reconfiguration (single rank): t[r] (+)= < y[r] , g[r] > over sites and channels, in double
__global__ __launch_bounds__(256) void k_cgen_pairdot(const float* __restrict__ y, const float* __restrict__ gm, int N,
                                                      int F, int Fp, double* __restrict__ t, int first) {
  __shared__ double s_w[4];
  const int r = blockIdx.x;
  const long long base = (long long)r * N * Fp;
  const int q = N * Fp / 4, fq = Fp / 4;
  double s = 0.0;
  for (int i = threadIdx.x; i < q; i += 256) {
    const f32x4 a = *(const f32x4*)(y + base + 4 * i), b = *(const f32x4*)(gm + base + 4 * i);
    const int c0 = 4 * (i % fq);
#pragma unroll
    for (int e = 0; e < 4; ++e) s += c0 + e < F ? (double)a[e] * (double)b[e] : 0.0;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double v = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
    t[r] = first ? v : t[r] + v;
  }
}

__global__ void k_cgen_tstore(const double* __restrict__ td, int rows, float* __restrict__ t) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows) t[i] = (float)td[i];
}

// centre[0] = c = mean of t (double sum, fixed order, one workgroup); usum[0] = sum_b (t_b - c): what k_sr_q reads as u[P]
__global__ __launch_bounds__(1024) void k_cgen_tmean(const float* __restrict__ t, int n, float* __restrict__ centre,
                                                     float* __restrict__ usum) {
  __shared__ double s[1024];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) a += (double)t[i];
  s[threadIdx.x] = a;
  __syncthreads();
  for (int d = 512; d >= 1; d >>= 1) {
    if ((int)threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float c = (float)(s[0] / (double)n);
    centre[0] = c;
    usum[0] = (float)(s[0] - (double)n * (double)c);
  }
}

// sharded solve: centre[0] = (sum of t over ALL ranks) / (samples of all ranks); usum[0] <- this rank's sum_b (t_b - c)
__global__ __launch_bounds__(1024) void k_cgen_tcentre_global(const float* __restrict__ t, int n, const float* __restrict__ count,
                                                              float* __restrict__ centre, float* __restrict__ usum) {
  __shared__ double s[1024];
  const float c = usum[0] / count[0];               // usum holds the all-reduced sum of t on entry
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) a += (double)t[i] - (double)c;
  s[threadIdx.x] = a;
  __syncthreads();
  for (int d = 512; d >= 1; d >>= 1) {
    if ((int)threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) { centre[0] = c; usum[0] = (float)s[0]; }
}

// per-position weights t_b - c
__global__ void k_cgen_wpos_centred(const float* __restrict__ t, const float* __restrict__ centre, long long row0, int rows,
                                    int N, float* __restrict__ wpos) {
  const float c = centre[0];
  const long long total = (long long)rows * N;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    wpos[i] = t[row0 + i / N] - c;
}

int cg_blocks(long long n) { const long long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b < 16384 ? b : 16384)); }

}  // namespace

hipError_t launch_cgen_im2col(hipStream_t s, const CgenIm2colArgs& a) {
  if (a.rows <= 0) return hipSuccess;
  const ConvGeom& g = a.g;
  const long long T = (long long)g.K * g.KW;
  if (a.layer == 0) {
    hipLaunchKernelGGL(k_cgen_im2col0, dim3(cg_blocks((long long)a.rows * g.N * T)), dim3(256), 0, s, a);
  } else {
    const int vec = g.F % 4 == 0 ? 4 : 1;
    const long long items = T * (g.F / vec), M = (long long)a.rows * g.N;
    const int ppb = (int)(items >= 512 ? 1 : (512 + items - 1) / items);
    const long long blocks = (M + ppb - 1) / ppb;
    const dim3 grid((unsigned)(blocks < 65536 ? blocks : 65536));
    if (vec == 4) hipLaunchKernelGGL(k_cgen_im2col<4>, grid, dim3(256), 0, s, a, ppb);
    else hipLaunchKernelGGL(k_cgen_im2col<1>, grid, dim3(256), 0, s, a, ppb);
  }
  return hipGetLastError();
}

hipError_t launch_cgen_rowsum(hipStream_t s, const float* fm, int rows, int N, int F, int Fp, double* out) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_cgen_rowsum, dim3(rows), dim3(256), 0, s, fm, rows, N, F, Fp, out);
  return hipGetLastError();
}

hipError_t launch_cgen_accept(hipStream_t s, float* configs, float* logit, const float* lnew, const int* iup,
                              const int* idn, const float* u, int B, int N, int oact, unsigned long long* accepted,
                              unsigned char* acc_mask) {
  hipLaunchKernelGGL(k_cgen_accept, dim3((B + 255) / 256), dim3(256), 0, s, configs, logit, lnew, iup, idn, u, B, N,
                     oact, accepted, acc_mask);
  return hipGetLastError();
}

hipError_t launch_cgen_step_tail(hipStream_t s, const float* fm, int N, int F, int Fp, float* configs, float* logit, int B,
                                 int oact, int* iup, int* idn, float* u, unsigned long long* accepted, uint32_t seed_lo,
                                 uint32_t seed_hi, int chain_offset, unsigned long long next_step, bool do_propose) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_cgen_step_tail, dim3(B), dim3(256), 0, s, fm, N, F, Fp, configs, logit, oact, iup, idn, u, accepted,
                     seed_lo, seed_hi, chain_offset, next_step, do_propose ? 1 : 0);
  return hipGetLastError();
}

hipError_t launch_cgen_fill(hipStream_t s, float* gm, const float* oscale, long long row0, int rows, int N, int F, int Fp) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_cgen_fill, dim3(cg_blocks((long long)rows * N * Fp)), dim3(256), 0, s, gm, oscale, row0, rows, N, F, Fp);
  return hipGetLastError();
}

hipError_t launch_cgen_dact(hipStream_t s, const float* d, const float* z, int pre, bool from_act, long long n, int F, int Fp,
                            float* out) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_cgen_dact, dim3(cg_blocks(n)), dim3(256), 0, s, d, z, pre, from_act ? 1 : 0, n, F, Fp, out);
  return hipGetLastError();
}

hipError_t launch_cgen_wpos(hipStream_t s, const float* w, long long row0, int rows, int N, float* wpos) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_cgen_wpos, dim3(cg_blocks((long long)rows * N)), dim3(256), 0, s, w, row0, rows, N, wpos);
  return hipGetLastError();
}

hipError_t launch_cgen_pack_t(hipStream_t s, const float* w, int T, int F, float* wt) {
  hipLaunchKernelGGL(k_cgen_pack_t, dim3(cg_blocks((long long)T * F * F)), dim3(256), 0, s, w, T, F, wt);
  return hipGetLastError();
}

hipError_t launch_cgen_pairdot(hipStream_t s, const float* y, const float* gm, int rows, int N, int F, int Fp, double* t,
                               bool first) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_cgen_pairdot, dim3(rows), dim3(256), 0, s, y, gm, N, F, Fp, t, first ? 1 : 0);
  return hipGetLastError();
}

hipError_t launch_cgen_tstore(hipStream_t s, const double* td, int rows, float* t) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_cgen_tstore, dim3((rows + 255) / 256), dim3(256), 0, s, td, rows, t);
  return hipGetLastError();
}

hipError_t launch_cgen_tmean(hipStream_t s, const float* t, int n, float* centre, float* usum) {
  hipLaunchKernelGGL(k_cgen_tmean, dim3(1), dim3(1024), 0, s, t, n, centre, usum);
  return hipGetLastError();
}

hipError_t launch_cgen_wpos_centred(hipStream_t s, const float* t, const float* centre, long long row0, int rows, int N,
                                    float* wpos) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_cgen_wpos_centred, dim3(cg_blocks((long long)rows * N)), dim3(256), 0, s, t, centre, row0, rows, N, wpos);
  return hipGetLastError();
}

hipError_t launch_cgen_tcentre_global(hipStream_t s, const float* t, int n, const float* count, float* centre, float* usum) {
  hipLaunchKernelGGL(k_cgen_tcentre_global, dim3(1), dim3(1024), 0, s, t, n, count, centre, usum);
  return hipGetLastError();
}
