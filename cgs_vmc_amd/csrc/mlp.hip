// Amplitude kernels of the fully-connected ansatz (wavefunctions.py:328-371) for gfx950.
//
//   k_pack      re-packs the flat parameter vector into MFMA-fragment-major weight images
//   (the first layer on raw +-1 configurations, z1 = x W1 + b1, runs through the LDS-tiled
//    MFMA GEMM of grad.hip; see first_layer() in vmc_api.hip)
//   k_tail16    layers 2..L + output dot for a list of rows, each row = cached z1 of a
//               chain (+ optional rank-2 exchange update), v_mfma_f32_16x16x4_f32,
//               activations never leave registers (transposed formulation, see below)
//   k_sweep16   persistent Metropolis exchange sampler: n_steps x mc_step
//               (graph_builders.py:38-89) in one launch, v_mfma_f32_16x16x4_f32
//
// Transposed formulation.  For a tile of samples j and hidden units i the kernels compute
// Y^T = W^T X^T, i.e. the MFMA A operand is a weight fragment (A[i][k] = W[k][i]) and the
// B operand is the activation fragment (B[k][j] = X[j][k]).  The 16x16 result then has the
// sample on the lane (col = lane&15) and the hidden unit on the register
// (row = 4(lane>>4) + r), which is exactly the B-operand shape of the next layer if its k index
// is visited in the order k(r, g) = 4g + r: accumulator register r of input tile ti IS the B
// operand of k-step r.  The weight image (p16) is stored in that k order, fragment-major, so
// every A-operand load is one fully coalesced 1 KiB dwordx4 wave load.
#include "common.hpp"
#include <cstdlib>
#include <type_traits>


// ------------------------------------------------------------------------------------ pack
__global__ void k_pack(const float* __restrict__ theta, int N, int H, int Hp, ParamLayout lay,
                       float* __restrict__ w1p, float* __restrict__ b1p, float* __restrict__ bh,
                       float* __restrict__ p16, float* __restrict__ p16t,
                       float* __restrict__ woutp, float* __restrict__ bout,
                       float* __restrict__ won) {
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long off_h0 = lay.off_h0;
  const long long per_h = (long long)H * H + H;
  const int n_hh = lay.n_hh;
  const int NT = Hp / 16;
  for (long long i = tid; i < (long long)N * Hp; i += stride) {
    const int n = (int)(i / Hp), c = (int)(i % Hp);
    w1p[i] = c < H ? theta[lay.off_w1 + (long long)n * H + c] : 0.f;
  }
  for (long long i = tid; i < Hp; i += stride) {
    b1p[i] = i < H ? theta[lay.off_b1 + i] : 0.f;
    // RBM: the output "dot" is a plain sum over the H log cosh values
    woutp[i] = i < H ? (lay.off_wout >= 0 ? theta[lay.off_wout + i] : 1.f) : 0.f;
    if (i == 0) bout[0] = theta[lay.off_bout];
  }
  if (lay.off_won >= 0)
    for (long long i = tid; i < N; i += stride) won[i] = theta[lay.off_won + i];
  for (long long i = tid; i < (long long)n_hh * Hp; i += stride) {
    const int l = (int)(i / Hp), c = (int)(i % Hp);
    bh[i] = c < H ? theta[off_h0 + l * per_h + (long long)H * H + c] : 0.f;
  }
  const long long n32 = (long long)n_hh * Hp * Hp;
  for (long long i = tid; i < n32; i += stride) {
    long long r = i;
    const int e = (int)(r & 3); r >>= 2;
    const int lane = (int)(r & 63); r >>= 6;
    const int ti = (int)(r % NT); r /= NT;
    const int to = (int)(r % NT); r /= NT;
    const int l = (int)r;
    const int k = 16 * ti + 4 * (lane >> 4) + e;
    const int n = 16 * to + (lane & 15);
    p16[i] = (k < H && n < H) ? theta[off_h0 + l * per_h + (long long)k * H + n] : 0.f;
    // transposed image for the back-propagation chain: A[i = n][k] = W[n][k]
    p16t[i] = (k < H && n < H) ? theta[off_h0 + l * per_h + (long long)n * H + k] : 0.f;
  }
}

hipError_t launch_pack(hipStream_t s, const float* theta, int N, int H, int Hp,
                       const ParamLayout& lay, float* w1p, float* b1p, float* bh, float* p16,
                       float* p16t, float* woutp, float* bout, float* won) {
  hipLaunchKernelGGL(k_pack, dim3(512), dim3(256), 0, s, theta, N, H, Hp, lay, w1p, b1p, bh, p16,
                     p16t, woutp, bout, won);
  return hipGetLastError();
}

// onsite term of the RBM ansatz (wavefunctions.py:436): out[r] = x_r . w_on, one wave per row
// (the bias b_on travels as `bout`)
__global__ __launch_bounds__(256) void k_onsite(const float* __restrict__ configs,
                                                const float* __restrict__ won, int rows, int N,
                                                float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float s = 0.f;
  for (int n = lane; n < N; n += 64) s = fmaf(configs[(long long)r * N + n], won[n], s);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  if (lane == 0) out[r] = s;
}

hipError_t launch_onsite(hipStream_t s, const float* configs, const float* won, int rows, int N,
                         float* out) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_onsite, dim3((rows + 3) / 4), dim3(256), 0, s, configs, won, rows, N, out);
  return hipGetLastError();
}

__global__ void k_iota_rows(int2* __restrict__ dst, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    dst[i] = make_int2(i, 0);
}

hipError_t launch_iota_rows(hipStream_t s, int2* dst, int n) {
  if (n <= 0) return hipSuccess;
  const int blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_iota_rows, dim3(blocks < 1024 ? blocks : 1024), dim3(256), 0, s, dst, n);
  return hipGetLastError();
}

// ------------------------------------------------------------------- per-activation dispatch
// act_tail.hip / act_sweep.hip are compiled once per hidden activation id; the launchers below
// pick the instantiation (an unknown id is an error, never a silent relu).
#define VMC_DECL_ACT(N)                                                                          \
  hipError_t launch_tail_inst_##N(hipStream_t, const TailArgs&, int, bool, bool);                \
  hipError_t launch_tail_lds_inst_##N(hipStream_t, const TailArgs&, int, bool, bool);            \
  hipError_t launch_backprop16_inst_##N(hipStream_t, const float*, float*, const float*,         \
                                        const float*, int, int, int, bool, const float*,         \
                                        const float*, const ElocFold&, const OutLayerSums&);     \
  hipError_t launch_sweep16_inst_##N(hipStream_t, const SweepArgs&, int);
VMC_DECL_ACT(0) VMC_DECL_ACT(1) VMC_DECL_ACT(2) VMC_DECL_ACT(3) VMC_DECL_ACT(4) VMC_DECL_ACT(5) VMC_DECL_ACT(6)
#undef VMC_DECL_ACT

#define VMC_ACT_SWITCH(act, CALL)                      \
  switch (act) {                                       \
    case 0: return CALL(0); case 1: return CALL(1);    \
    case 2: return CALL(2); case 3: return CALL(3);    \
    case 4: return CALL(4); case 5: return CALL(5);    \
    case 6: return CALL(6);                            \
    default: return hipErrorInvalidValue;              \
  }

hipError_t launch_tail(hipStream_t s, const TailArgs& a, int Hp, bool ratio_mode, bool rbm) {
#define CALL(N) launch_tail_inst_##N(s, a, Hp, ratio_mode, rbm)
  VMC_ACT_SWITCH(a.act, CALL)
#undef CALL
}

// 384 or 512 padded units and at least one H x H layer (the sampler's LDS need is checked by vmc_create);
// LDS of k_tail_lds: two operand buffers [2 halves][NT][64][4], partial dots, row meta, biases, w_out
bool tail_lds_supported(int Hp, int n_hidden) { return plan_tail_lds_supported(Hp, n_hidden); }

hipError_t launch_tail_lds(hipStream_t s, const TailArgs& a, int Hp, bool ratio_mode, bool rbm) {
#define CALL(N) launch_tail_lds_inst_##N(s, a, Hp, ratio_mode, rbm)
  VMC_ACT_SWITCH(a.act, CALL)
#undef CALL
}

hipError_t launch_backprop16(hipStream_t s, const float* act_all, float* delta_all,
                             const float* p16t, const float* woutp, int B, int Hp, int n_hidden,
                             bool rbm, int act, const float* dact_all, const float* oscale,
                             const ElocFold& eloc, const OutLayerSums& out) {
#define CALL(N) launch_backprop16_inst_##N(s, act_all, delta_all, p16t, woutp, B, Hp, n_hidden, rbm, dact_all, oscale, eloc, out)
  VMC_ACT_SWITCH(act, CALL)
#undef CALL
}

hipError_t launch_sweep16(hipStream_t s, const SweepArgs& a, int Hp) {
  if (a.B <= 0) return hipSuccess;
#define CALL(N) launch_sweep16_inst_##N(s, a, Hp)
  VMC_ACT_SWITCH(a.act, CALL)
#undef CALL
}

// LDS the sampler needs at least (W1 streamed from L2); vmc_create rejects shapes beyond 160 KiB
size_t sweep_lds_required(int N, int Hp, int n_hidden, bool rbm) { return plan_sweep_lds_required(N, Hp, n_hidden, rbm); }
