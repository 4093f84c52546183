// Micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 on gfx950 for
// NCHAIN independent accumulator chains per wave and WAVES waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_peak.hip -o gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NCHAIN, bool BIG>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NCHAIN];
  f32x4 acc4[NCHAIN];
#pragma unroll
  for (int c = 0; c < NCHAIN; ++c) { for (int r = 0; r < 16; ++r) acc[c][r] = 0.f; for (int r = 0; r < 4; ++r) acc4[c][r] = 0.f; }
  float a = a0 + threadIdx.x, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < NCHAIN; ++c) {
        if (BIG) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        else acc4[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[c], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCHAIN; ++c) { for (int r = 0; r < 16; ++r) s += acc[c][r]; for (int r = 0; r < 4; ++r) s += acc4[c][r]; }
  if (s == 12345.f) out[threadIdx.x] = s;
}

template <int NCHAIN, bool BIG>
void run(int waves_per_simd, float* d) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const dim3 grid(256), block(256 * waves_per_simd);
  hipLaunchKernelGGL((k<NCHAIN, BIG>), grid, block, 0, 0, d, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NCHAIN, BIG>), grid, block, 0, 0, d, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)iters * 16 * NCHAIN * 256 * 4 * waves_per_simd;
  const double flops = mfmas * (BIG ? 4096.0 : 2048.0);
  printf("%s chains=%d waves/SIMD=%d : %.3f ms  %.1f TFLOP/s  (%.1f%% of 157.3)\n", BIG ? "32x32x2" : "16x16x4",
         NCHAIN, waves_per_simd, ms, flops / ms / 1e9, 100.0 * flops / ms / 1e9 / 157.3);
}

int main() {
  float* d; hipMalloc(&d, 4096);
  run<1, true>(1, d); run<2, true>(1, d); run<4, true>(1, d); run<1, true>(2, d); run<2, true>(2, d);
  run<1, false>(1, d); run<2, false>(1, d); run<4, false>(1, d); run<1, false>(2, d); run<2, false>(2, d);
  return 0;
}
