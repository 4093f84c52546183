// Convolutional ansatz types on gfx950: host side (launchers, LDS budgets, parameter packing, the
// weight-gradient reduction) and the kernels for up to 16 filters (NCB = 1).  The kernels are the
// templates of conv_kernels.hpp; conv32 / conv48 / conv64.hip instantiate them for 17 .. 64 filters (NCB = 2, 3, 4).
#include "conv_kernels.hpp"

// NCB = 2, 3, 4 launchers (conv32.hip, conv48.hip, conv64.hip through conv_wide.hpp)
#define CONV_DECLARE_CB(N)                                                                              \
  hipError_t conv_launch_rows_cb##N(hipStream_t s, const ConvRowsArgs& a, dim3 grid, size_t lds);       \
  hipError_t conv_launch_sweep_cb##N(hipStream_t s, const ConvSweepArgs& a, dim3 grid, size_t lds);     \
  hipError_t conv_launch_back_cb##N(hipStream_t s, const ConvBackArgs& a, dim3 grid, size_t lds);       \
  hipError_t conv_launch_dw_cb##N(hipStream_t s, const ConvDwArgs& a, dim3 grid, size_t lds);           \
  hipError_t conv_launch_sr_rowdot_cb##N(hipStream_t s, const ConvSrRowdotArgs& a, dim3 grid, size_t lds);
CONV_DECLARE_CB(2)
CONV_DECLARE_CB(3)
CONV_DECLARE_CB(4)
// the instantiation for a.g.NCB channel blocks (vmc_create admits 1 .. CONV_MAX_NCB)
#define CONV_BY_NCB(NAME, ...)                                         \
  switch (a.g.NCB) {                                                   \
    case 1: return conv_launch_##NAME##_t<1>(__VA_ARGS__);             \
    case 2: return conv_launch_##NAME##_cb2(__VA_ARGS__);              \
    case 3: return conv_launch_##NAME##_cb3(__VA_ARGS__);              \
    case 4: return conv_launch_##NAME##_cb4(__VA_ARGS__);              \
    default: return hipErrorInvalidValue;                              \
  }
static_assert(CONV_MAX_NCB == 4, "one launcher set per channel-block count");

namespace {

// g1 / g2 += sum over slices (fixed order) of the partial sums, scattered to the theta layout:
// convolution l: w [K][K][cin][F] then b [F]   (snt.Conv2D variable order).  Grid (64 parameters,
// layer); the four waves of a workgroup take the slices s, s + 4, ... and their four partial sums are
// added as (0 + 1) + (2 + 3).  g1 == nullptr: the weighted sum only (the other half of ws is unwritten)
__global__ __launch_bounds__(256) void k_conv_dw_reduce(ConvDwArgs a, int KK) {
  __shared__ float p1[4][64], p2[4][64];
  const ConvGeom& g = a.g;
  const int CW = 16 * g.NCB;
  const size_t rows = (size_t)KK * CW + 1;
  const int l = blockIdx.y;
  const long long p0 = (long long)KK * g.F + g.F, pl = (long long)KK * g.F * g.F + g.F;
  const long long off = l == 0 ? 0 : p0 + (long long)(l - 1) * pl;
  const int cin = l == 0 ? 1 : g.F;
  const long long nw = (long long)KK * cin * g.F, np = nw + g.F;
  const int il = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const long long i = (long long)blockIdx.x * 64 + il;
  float s1 = 0.f, s2 = 0.f;
  if (i < np) {
    size_t src;
    if (i < nw) {
      const int co = (int)(i % g.F);
      const int ci = (int)((i / g.F) % cin);
      const int tap = (int)(i / ((long long)g.F * cin));
      src = ((size_t)tap * CW + ci) * CW + co;
    } else {
      src = (size_t)KK * CW * CW + (size_t)(i - nw);
    }
    for (int sl = sg; sl < a.n_slices; sl += 4) {
      const float* w1 = a.ws + (((size_t)sl * g.n_conv + l) * 2) * rows * CW;
      if (a.g1) s1 += w1[src];
      s2 += w1[rows * CW + src];
    }
  }
  p1[sg][il] = s1; p2[sg][il] = s2;
  __syncthreads();
  if (sg == 0 && i < np) {
    if (a.g1) a.g1[off + i] += (p1[0][il] + p1[1][il]) + (p1[2][il] + p1[3][il]);
    a.g2[off + i] += (p2[0][il] + p2[1][il]) + (p2[2][il] + p2[3][il]);
  }
}

// theta (snt.Conv2D order: per convolution w[K][K][cin][F], b[F]) -> fragment images (conv.hpp)
__global__ void k_conv_pack(const float* __restrict__ theta, ConvGeom g, float* w0, float* wf,
                            float* wb, float* bias) {
  const int KK = g.K * g.KW, Q0 = (KK + 3) / 4, NCB = g.NCB;
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long p0 = (long long)KK * g.F + g.F;                 // parameters of convolution 0
  const long long pl = (long long)KK * g.F * g.F + g.F;           // of every later one
  for (long long i = tid; i < (long long)NCB * Q0 * 64; i += stride) {
    const int cb = (int)(i / (Q0 * 64)), q = (int)((i / 64) % Q0), lane = (int)(i % 64), m = lane & 15, gq = lane >> 4;
    const int tap = 4 * q + gq, co = 16 * cb + m;
    w0[i] = (tap < KK && co < g.F) ? theta[(long long)tap * g.F + co] : 0.f;
  }
  for (long long i = tid; i < (long long)g.n_conv * 16 * NCB; i += stride) {
    const int l = (int)(i / (16 * NCB)), c = (int)(i % (16 * NCB));
    const long long base = l == 0 ? (long long)KK * g.F : p0 + (long long)(l - 1) * pl + (long long)KK * g.F * g.F;
    bias[i] = c < g.F ? theta[base + c] : 0.f;
  }
  const long long per_block = (long long)KK * 256, per_layer = per_block * NCB * NCB;
  for (long long i = tid; i < (long long)(g.n_conv - 1) * per_layer; i += stride) {
    const int l = (int)(i / per_layer) + 1;
    const long long rl = i % per_layer;
    const int blk = (int)(rl / per_block), bo = blk / NCB, bi = blk % NCB;   // image block (output, input)
    const long long r = rl % per_block;
    const int tap = (int)(r / 256), lane = (int)((r % 256) / 4), e = (int)(r % 4);
    const int m = lane & 15, gq = lane >> 4;
    const float* w = theta + p0 + (long long)(l - 1) * pl;         // [tap][cin][cout]
    // forward: output channel (lane m) = cout of block bo, k index = cin of block bi
    const int ci = 16 * bi + 4 * gq + e, co = 16 * bo + m;
    wf[i] = (ci < g.F && co < g.F) ? w[((long long)tap * g.F + ci) * g.F + co] : 0.f;
    // transposed convolution: output channel (lane m) = cin of block bo, k index = cout of block bi, taps flipped
    const int ci2 = 16 * bo + m, co2 = 16 * bi + 4 * gq + e, tap2 = KK - 1 - tap;
    wb[i] = (ci2 < g.F && co2 < g.F) ? w[((long long)tap2 * g.F + ci2) * g.F + co2] : 0.f;
  }
}

}  // namespace

// the planners live in plan.hpp; these are the names the rest of the library calls
static bool conv_one_wg_per_cu() {     // experiment knob: one workgroup per CU with twice the samples
  static const int v = getenv("CGS_VMC_CONV_WG_PER_CU") ? atoi(getenv("CGS_VMC_CONV_WG_PER_CU")) == 1 : 0;
  return v != 0;
}
size_t conv_lds_cap(const ConvGeom& g) { return plan_conv_lds_cap(g, conv_one_wg_per_cu()); }
int conv_waves(const ConvGeom& g) { return plan_conv_waves(g); }
size_t conv_rows_lds(const ConvGeom& g, int G) { return plan_conv_rows_lds(g, G); }
int conv_pick_group(const ConvGeom& g, int waves) { return plan_conv_pick_group(g, waves, conv_one_wg_per_cu()); }
int conv_pick_sweep_group(const ConvGeom& g, long long B, int num_cus, int waves) {
  return plan_conv_pick_sweep_group(g, B, num_cus, waves, conv_one_wg_per_cu());
}
long long conv_num_params(int n_conv, int F, int taps) { return plan_num_params_conv(n_conv, F, taps); }

hipError_t launch_conv_pack(hipStream_t s, const float* theta, const ConvGeom& g, float* w0,
                            float* wf, float* wb, float* bias) {
  hipLaunchKernelGGL(k_conv_pack, dim3(64), dim3(256), 0, s, theta, g, w0, wf, wb, bias);
  return hipGetLastError();
}

hipError_t launch_conv_rows(hipStream_t s, const ConvRowsArgs& a, int num_cus) {
  if (a.n_rows <= 0) return hipSuccess;
  const size_t lds = conv_rows_lds(a.g, a.G);
  const dim3 grid(plan_conv_grid(a.g, a.n_rows, a.G, num_cus));      // one workgroup per co-resident slot
  CONV_BY_NCB(rows, s, a, grid, lds);
}

hipError_t launch_conv_sweep(hipStream_t s, const ConvSweepArgs& a) {
  const dim3 grid((a.B + a.G - 1) / a.G);
  const size_t lds = conv_rows_lds(a.g, a.G);
  CONV_BY_NCB(sweep, s, a, grid, lds);
}

hipError_t launch_conv_back(hipStream_t s, const ConvBackArgs& a, int num_cus) {
  const size_t lds = conv_rows_lds(a.g, a.G);
  const dim3 grid(plan_conv_grid(a.g, a.B, a.G, num_cus));
  CONV_BY_NCB(back, s, a, grid, lds);
}

hipError_t launch_conv_dw(hipStream_t s, const ConvDwArgs& a_in) {
  ConvDwArgs a = a_in;
  const dim3 grid(a.n_slices, a.g.n_conv, plan_conv_dw_grid_z(a.g));
  // the whole sample at once when it fits (two workgroups per CU at 16 filters when THAT fits), else
  // the largest band of rows that does
  const char* e_band = getenv("CGS_VMC_CONV_DW_BAND");      // test knob: force bands of this many rows
  a.band_rows = plan_conv_dw_band(a.g, e_band ? atoi(e_band) : 0);
  if (a.band_rows < 1) return hipErrorInvalidValue;
  const size_t lds = plan_conv_dw_lds(a.g, a.band_rows);
  hipError_t e = [&]() -> hipError_t { CONV_BY_NCB(dw, s, a, grid, lds); }();
  if (e != hipSuccess) return e;
  const long long np_max = (long long)a.g.K * a.g.KW * a.g.F * a.g.F + a.g.F;
  hipLaunchKernelGGL(k_conv_dw_reduce, dim3((unsigned)((np_max + 63) / 64), a.g.n_conv), dim3(256), 0, s, a,
                     a.g.K * a.g.KW);
  return hipGetLastError();
}

hipError_t launch_conv_sr_rowdot(hipStream_t s, const ConvSrRowdotArgs& a, int num_cus) {
  if (a.n_rows <= 0) return hipSuccess;
  const size_t lds = conv_rows_lds(a.g, a.G);
  const dim3 grid(plan_conv_grid(a.g, a.n_rows, a.G, num_cus));
  CONV_BY_NCB(sr_rowdot, s, a, grid, lds);
}
