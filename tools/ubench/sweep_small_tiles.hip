// Micro-benchmark (round 5): the weight-traffic skeleton of a config-5 sampler with 4- or 8-chain
// tiles on the 4x4 MFMA shape, before building it.  Config 5's shard (1,024 chains, FC 6 x 256) gives the 16-chain
// sampler 64 tiles = 64 of 256 CUs; with CH chains per workgroup 1024 / CH CUs work, but every CU then streams ALL
// five H x H layers per mc_step: 1.25 MiB of fp32 fragments, of which 224 KiB stay in registers (as in k_sweep16).
// v_mfma_f32_4x4x1_16B_f32: 16 blocks of 4 output units x 4 chains, k = 1: one weight VGPR (64 output units) per
// 512 flops and 8 cycles; an 8-chain tile uses each weight VGPR twice.
// Skeleton: 8 waves (wave w owns 32 of the 256 output units = half an MFMA's 64... two waves share a fragment's
// halves; simplified here: every wave streams 1/8 of the image), 4-deep ring of 1 KiB fragments, two barriers per
// layer.  Prints us per mc_step for grids of 256 (4 chains) and 128 (8 chains) workgroups and the streamed bytes
// per clock and CU.  k_sweep16 today: 22.9 us per mc_step at config 5 (5.86 ms per 256-step sweep) on 64 CUs.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/sweep_small_tiles.hip -o /tmp/sst && /tmp/sst
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// image: [layer 5][wave 8][item][64 lanes][4 floats]; an item = one f32x4 per lane = 4 weight VGPRs
template <int CHG>   // chain groups of 4: 1 (4-chain tile) or 2 (8-chain tile)
__global__ __launch_bounds__(512) void k(const float* __restrict__ w, int steps, int items_per_layer, int resident_items, float* sink) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  typedef const __attribute__((address_space(1))) f32x4* gp;
  const float* wbase = w;
  f32x4 acc[CHG];
  for (int c = 0; c < CHG; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float x0 = 1.0f + lane, x1 = 2.0f + lane;
  f32x4 ring[4];
  const int n_layers = 5;
  auto frag = [&](int l, int it) { return ((gp)(wbase + (((long long)l * 8 + wave) * items_per_layer + it) * 256) + lane)[0]; };
  for (int s = 0; s < steps; ++s) {
    asm volatile("" : "+s"(wbase));
    for (int l = 0; l < n_layers; ++l) {
      const int first = l == 0 ? resident_items : 0;       // layer 0's first items stand for the register-resident part
#pragma unroll
      for (int q = 0; q < 3; ++q) ring[q] = frag(l, first + q);
      for (int it = first; it < items_per_layer; it += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int nx = it + u + 3;
          ring[(u + 3) & 3] = frag(l, nx < items_per_layer ? nx : first);
          __builtin_amdgcn_sched_barrier(0);
          const f32x4 v = ring[u];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(v[e], x0, acc[0], 0, 0, 0);
            if (CHG > 1) acc[CHG - 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(v[e], x1, acc[CHG - 1], 0, 0, 0);
          }
        }
      }
      __syncthreads();
      __syncthreads();
    }
  }
  float sum = 0.f;
  for (int c = 0; c < CHG; ++c) sum += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  if (sum == 12345.f) sink[threadIdx.x] = sum;
}

template <int CHG>
void run(const float* w, float* sink, int grid) {
  const int steps = 200;
  // per layer and wave: 256 x 256 floats / 8 waves / 256 floats per item = 32 items; resident: 224 KiB / 8 waves / 1 KiB = 28 items of layer 0
  const int items = 32, resident = 28;
  hipLaunchKernelGGL((k<CHG>), dim3(grid), dim3(512), 0, 0, w, 10, items, resident, sink);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<CHG>), dim3(grid), dim3(512), 0, 0, w, steps, items, resident, sink);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / steps;
  const double kib = 5 * 256.0 - 28 * 8.0;
  printf("%d-chain tiles, %3d workgroups: %.2f us per mc_step -> %.2f ms per 256-step sweep; streamed %.0f KiB per step and CU = %.1f B per 2.3 GHz clock and CU, %.1f TB/s chip\n",
         4 * CHG, grid, us, us * 256e-3, kib, kib * 1024 / (us * 2300.0), kib * 1024 * grid / (us * 1e-6) / 1e12);
}

int main() {
  float *w, *sink;
  const size_t bytes = (size_t)5 * 256 * 256 * 4;
  (void)hipMalloc(&w, bytes); (void)hipMalloc(&sink, 4096);
  std::vector<float> h(bytes / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 1023) * 1e-3f - 0.5f;
  (void)hipMemcpy(w, h.data(), bytes, hipMemcpyHostToDevice);
  printf("k_sweep16 at config 5 today: 22.9 us per mc_step (5.86 ms per sweep) on 64 of 256 CUs\n");
  run<1>(w, sink, 256); run<2>(w, sink, 128); run<1>(w, sink, 64); run<2>(w, sink, 64);
  return 0;
}
