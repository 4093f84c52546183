// Micro-benchmark: the k_tail loop skeleton -- one 1 KiB dwordx4 wave load per 256 MFMA cycles
// through an 8-deep register ring, one wave per SIMD -- against the size of the streamed
// weight image (L1 / L2 resident) and the MFMA shape.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_stream.hip -o tools/ubench/mfma_stream
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool BIG, int RD>
__global__ __launch_bounds__(256) void k(const float* __restrict__ w, int n_items, int iters,
                                        float* out, float b0) {
  const int lane = threadIdx.x & 63;
  const f32x4* wp = (const f32x4*)w + lane;
  f32x16 acc; f32x4 a0, a1;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int r = 0; r < 4; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
  f32x4 ring[RD];
#pragma unroll
  for (int i = 0; i < RD - 1; ++i) ring[i] = wp[i * 64];
  const float b = b0 + lane;
  int item = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < RD; ++u) {
      int nxt = item + RD - 1; if (nxt >= n_items) nxt -= n_items;
      ring[(u + RD - 1) % RD] = wp[nxt * 64];
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 v = ring[u];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (BIG) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[e], b, acc, 0, 0, 0);
        else {
          a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[e], b, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[e], b + 1.f, a1, 0, 0, 0);
        }
      }
      item = item + 1 == n_items ? 0 : item + 1;
    }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  for (int r = 0; r < 4; ++r) s += a0[r] + a1[r];
  if (s == 12345.f) out[threadIdx.x] = s;
}

template <bool BIG, int RD>
void run(const float* w, int kb, float* d) {
  const int n_items = kb;   // 1 KiB per item
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<BIG, RD>), dim3(256), dim3(256), 0, 0, w, n_items, 10, d, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<BIG, RD>), dim3(256), dim3(256), 0, 0, w, n_items, iters, d, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)iters * RD * (BIG ? 4 * 4096.0 : 8 * 2048.0) * 256 * 4;
  printf("%s ring=%d image=%5d KiB : %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", BIG ? "32x32x2" : "16x16x4", RD, kb, ms,
         flops / ms / 1e9, 100.0 * flops / ms / 1e9 / 157.3);
}

int main() {
  float *w, *d; hipMalloc(&w, 8 << 20); hipMemset(w, 0, 8 << 20); hipMalloc(&d, 4096);
  for (int kb : {16, 256, 1024, 4096}) { run<true, 8>(w, kb, d); run<false, 8>(w, kb, d); }
  run<true, 16>(w, 1024, d); run<false, 16>(w, 1024, d);
  return 0;
}
