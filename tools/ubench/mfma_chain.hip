// Probe (round 6): issue interval of DEPENDENT v_mfma_f32_4x4x1_16B_f32 (one accumulator chain per wave, as in
// k_sweep8) against two independent chains, with 1 and 2 waves per SIMD.  Prints cycles per MFMA per wave.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_chain.hip -o tools/ubench/mfma_chain && tools/ubench/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void k(float* out, int iters, unsigned long long* cyc) {
  f32x4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < 32; ++q)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 3, 5, 2);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int CHAINS>
void run(int threads, float* out, unsigned long long* cyc) {
  const int iters = 1000;
  hipLaunchKernelGGL((k<CHAINS>), dim3(1), dim3(threads), 0, 0, out, 10, cyc);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<CHAINS>), dim3(1), dim3(threads), 0, 0, out, iters, cyc);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 32 * CHAINS;
  printf("%d chain(s) per wave, %d waves per SIMD: %.2f ns per MFMA per wave (%.1f clocks at 2.4 GHz); s_memtime ticks per MFMA %.2f\n",
         CHAINS, threads / 256, ms * 1e6 / n, ms * 1e6 / n * 2.4, (double)h / n);
}

int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 4096 * 4); (void)hipMalloc(&cyc, 8);
  run<1>(256, out, cyc); run<2>(256, out, cyc); run<1>(512, out, cyc); run<2>(512, out, cyc);
  return 0;
}
