// Micro-benchmark: does VALU work issued between v_mfma_f32_16x16x4_f32 instructions cost matrix
// pipe time on gfx950?  Two waves per SIMD; per pair of MFMAs each wave also issues NV independent
// VALU instructions of kind KIND (0: v_xor/v_add, 1: 32-bit integer multiply-add, 2: v_fma_f32).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu.hip -o gpurun_out/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a0, float b0, unsigned seed) {
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  float a = a0 + threadIdx.x, b = b0;
  unsigned x[4] = {seed + threadIdx.x, seed * 3u, seed * 5u + threadIdx.x, seed * 7u};
  float f[4] = {a0, b0, a0 * 2.f, b0 * 3.f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (KIND == 0) x[v & 3] ^= x[(v + 1) & 3] + 0x9E3779B9u;
        else if (KIND == 1) x[v & 3] = x[v & 3] * 0xD2511F53u + x[(v + 1) & 3];
        else f[v & 3] = fmaf(f[v & 3], 1.0001f, f[(v + 1) & 3]);
      }
      __builtin_amdgcn_sched_barrier(0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int r = 0; r < 4; ++r) s += acc0[r] + acc1[r] + f[r] + (float)x[r];
  if (s == 12345.f) out[threadIdx.x] = s;
}

template <int NV, int KIND>
void run(float* d) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const dim3 grid(256), block(512);
  hipLaunchKernelGGL((k<NV, KIND>), grid, block, 0, 0, d, 10, 1.f, 2.f, 7u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, KIND>), grid, block, 0, 0, d, iters, 1.f, 2.f, 7u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)iters * 32 * 256 * 8;
  printf("kind %d  VALU per 2 MFMAs = %d : %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", KIND, NV, ms,
         mfmas * 2048.0 / ms / 1e9, 100.0 * mfmas * 2048.0 / ms / 1e9 / 157.3);
}

int main() {
  float* d; hipMalloc(&d, 4096);
  run<0, 0>(d); run<1, 0>(d); run<2, 0>(d); run<4, 0>(d); run<8, 0>(d);
  run<1, 1>(d); run<2, 1>(d); run<4, 1>(d);
  run<2, 2>(d); run<4, 2>(d); run<8, 2>(d);
  return 0;
}
