// Micro-benchmark (round 5): the WEIGHT TRAFFIC SKELETON of a 3 x bf16 split sampler
// (k_sweep16s), before building it.  One workgroup of 8 waves (two per SIMD) per CU owns 16 chains; wave w owns
// 32 output units (two 16-unit tiles) of both 256 x 256 layers.  In split form a weight is three bf16 terms (6
// bytes instead of 4): 768 KiB per mc_step per CU, every byte wanted by exactly ONE wave (nothing to share through
// LDS), of which RES k-steps of layer 0 stay in registers (24 registers per k-step and wave).  The rest streams
// L2 -> registers through a PF-item ring, six 1 KiB fragments and twelve v_mfma_f32_16x16x32_bf16 per item and
// wave, all 256 CUs streaming the same image.  Two workgroup barriers per step stand for the layer hand-overs;
// the serial phases of a step (proposals, build, accept: ~4.5 k cycles in k_sweep16) are NOT in here.
//   prints cycles per step (s_memtime) and the streamed bytes per clock and CU, against k_sweep16's 17.9 k cycles
//   for the same two layers on the fp32 matrix cores (8.6 k + 9.3 k).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/sweep_split_stream.hip -o /tmp/sweep_split_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

struct Frag { u32x4 h, m, l; };

// image: [layer][wave][kt][to][term][64][4] dwords
template <int RES, int PF, bool MFMA>
__global__ __launch_bounds__(512) void k(const unsigned* __restrict__ w, int steps, unsigned long long* cyc, float* sink) {
  constexpr int KT = 8, TO = 2;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  typedef const __attribute__((address_space(1))) u32x4* gp;
  const unsigned* wbase = w;
  auto frag = [&](int l, int kt, int to) {
    gp p = (gp)(wbase + ((((long long)l * 8 + wave) * KT + kt) * TO + to) * 3 * 256) + lane;
    Frag f; f.h = p[0]; f.m = p[64]; f.l = p[128];
    return f;
  };
  Frag res[RES > 0 ? RES : 1][TO];
#pragma unroll
  for (int kt = 0; kt < RES; ++kt)
#pragma unroll
    for (int to = 0; to < TO; ++to) res[kt][to] = frag(0, kt, to);
  Frag ring[PF][TO];
  // stream order: (layer 0, kt = RES..7), (layer 1, kt = 0..7): NS items per step
  constexpr int NS = (KT - RES) + KT;
  auto item_l = [&](int q) { return q < KT - RES ? 0 : 1; };
  auto item_kt = [&](int q) { return q < KT - RES ? RES + q : q - (KT - RES); };
#pragma unroll
  for (int q = 0; q < PF - 1; ++q)
#pragma unroll
    for (int to = 0; to < TO; ++to) ring[q][to] = frag(item_l(q), item_kt(q), to);
  f32x4 acc[TO];
#pragma unroll
  for (int to = 0; to < TO; ++to) acc[to] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 xh = {0x3f803f80u + lane, 1, 2, 3}, xm = {4, 5, 6, 7u + lane}, xl = {8, 9u + lane, 10, 11};
  auto mult = [&](const Frag (&f)[TO]) {
    if (!MFMA) {
#pragma unroll
      for (int to = 0; to < TO; ++to) acc[to][0] += __uint_as_float(f[to].h[0] ^ f[to].m[1] ^ f[to].l[2]);
      return;
    }
#pragma unroll
    for (int to = 0; to < TO; ++to) {
      acc[to] = mfma(f[to].l, xh, acc[to]); acc[to] = mfma(f[to].h, xl, acc[to]); acc[to] = mfma(f[to].m, xm, acc[to]);
      acc[to] = mfma(f[to].m, xh, acc[to]); acc[to] = mfma(f[to].h, xm, acc[to]); acc[to] = mfma(f[to].h, xh, acc[to]);
    }
  };
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int s = 0; s < steps; ++s) {
    asm volatile("" : "+s"(wbase));
#pragma unroll
    for (int kt = 0; kt < RES; ++kt) mult(res[kt]);
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      const int nq = (q + PF - 1) % NS;                   // (wraps into the next step: the weights do not change)
#pragma unroll
      for (int to = 0; to < TO; ++to) ring[(q + PF - 1) % PF][to] = frag(item_l(nq), item_kt(nq), to);
      __builtin_amdgcn_sched_barrier(0);
      mult(ring[q % PF]);
      if (q == KT - RES - 1 || q == NS - 1) __syncthreads();   // the layer hand-overs
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  float sum = 0.f;
#pragma unroll
  for (int to = 0; to < TO; ++to) sum += acc[to][0] + acc[to][1] + acc[to][2] + acc[to][3];
  if (sum == 12345.f) sink[threadIdx.x] = sum;
}

template <int RES, int PF, bool MFMA>
void run(const unsigned* w, unsigned long long* dc, float* sink) {
  const int steps = 400, grid = 256;
  hipLaunchKernelGGL((k<RES, PF, MFMA>), dim3(grid), dim3(512), 0, 0, w, 20, dc, sink);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<RES, PF, MFMA>), dim3(grid), dim3(512), 0, 0, w, steps, dc, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(grid);
  hipMemcpy(h.data(), dc, grid * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += (double)v; mean /= grid;
  const double per_step = mean / steps;                    // s_memtime ticks at 100 MHz x ... : report both
  const double us_step = ms * 1e3 / steps;
  const double streamed_kb = (768.0 - RES * 8 * 2 * 3.0);  // KiB per step per CU (RES k-steps x 8 waves x 2 tiles x 3 KiB)
  printf("RES %d PF %d %s : %.2f us per step  (%.0f cycles at 2.3 GHz; counter %.0f ticks)  streamed %.0f KiB -> %.1f B per 2.3 GHz clock per CU, %.2f TB/s chip\n",
         RES, PF, MFMA ? "mfma " : "loads", us_step, us_step * 2300.0, per_step, streamed_kb, streamed_kb * 1024 / (us_step * 2300.0),
         streamed_kb * 1024 * 256 / (us_step * 1e-6) / 1e12);
}

int main() {
  unsigned* w; unsigned long long* dc; float* sink;
  hipMalloc(&w, 768 << 10); hipMalloc(&dc, 256 * 8); hipMalloc(&sink, 4096);
  std::vector<unsigned> h((768 << 10) / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c003c00u ^ (unsigned)(i * 2654435761u >> 9 & 0x007f007f);   // random small bf16 pairs
  hipMemcpy(w, h.data(), 768 << 10, hipMemcpyHostToDevice);
  printf("k_sweep16 (fp32 MFMA) spends 17.9 k cycles per mc_step in its two H x H layers; a split sampler would need <= ~9 k for a step <= 1.35 ms\n");
  run<4, 2, true>(w, dc, sink); run<4, 3, true>(w, dc, sink); run<2, 3, true>(w, dc, sink); run<0, 3, true>(w, dc, sink);
  run<4, 2, false>(w, dc, sink); run<4, 3, false>(w, dc, sink); run<0, 3, false>(w, dc, sink);
  return 0;
}
