// Micro-benchmark for VERDICT r2 item 3 (config-5 sampler on a 2-CU cooperative tile): what does ONE
// activation exchange between two workgroups on different CUs cost in the sampler's geometry?
//
// Model of the proposal: workgroups 2p and 2p+1 (512 threads = 8 waves, one per CU) own the same 16
// chains; per H x H layer each computes 128 of the 256 output units (128 instead of 256 MFMAs per
// SIMD) and hands its half of the layer's activations -- 16 chains x 128 units x 4 B = 8 KB -- to the
// partner through L2, which needs all 256 units as the next layer's B operand.
//
// One round = what a layer boundary would do:
//   every thread stores its 16 B of the outbox (512 x 16 B = 8 KB) with sc1 (write-through) stores,
//   s_waitcnt vmcnt(0), workgroup barrier, lane 0 stores the flag (= round) with sc1;
//   lane 0 polls the partner's flag with sc1 loads (s_sleep between polls), workgroup barrier,
//   every thread loads 16 B of the partner's outbox with an sc1 load.
// (the "EVERY store sc1, drained, flag; poll, barrier, EVERY load sc1" form of MI355X_MICROARCH.md,
// valid for hipMalloc memory with one workgroup per CU.)
// Variants: partner on the same XCD (blocks b and b + 8 under round-robin XCD dispatch) or on the
// next XCD (b and b + 1); chip otherwise idle, or every workgroup also streams WSTREAM KB of weights
// from L2 per round through its 8 waves (the sampler streams 288 KB per layer per CU).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/xcu_exchange.hip -o gpurun_out/xcu_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) f32x4* gf4_p;
typedef __attribute__((address_space(1))) unsigned* gu_p;

__device__ __forceinline__ void store_sc1(f32x4* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"((gf4_p)p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 load_sc1(const f32x4* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"((gf4_p)p) : "memory");
  return v;
}
__device__ __forceinline__ void store_flag(unsigned* p, unsigned v) {
  asm volatile("global_store_dword %0, %1, off sc1" ::"v"((gu_p)p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned load_flag(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"((gu_p)p) : "memory");
  return v;
}

// pair layout: partner(b) = b ^ stride-bit.  same_xcd: blocks b and b + 8 (b / 8 even <-> odd)
__global__ __launch_bounds__(512) void k_exchange(f32x4* box, unsigned* flags, const f32x4* weights,
                                                  int rounds, int same_xcd, int wstream_vec,
                                                  unsigned long long* cycles, float* sink, int exchange) {
  const int b = blockIdx.x, t = threadIdx.x;
  const int partner = same_xcd ? (((b >> 3) ^ 1) << 3 | (b & 7)) : (b ^ 1);
  f32x4* mine = box + (size_t)b * 512;
  const f32x4* theirs = box + (size_t)partner * 512;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  f32x4 val = {(float)t, 1.f, 2.f, 3.f};
  unsigned long long t0 = 0;
  for (int r = 1; r <= rounds; ++r) {
    if (r == 2 && t == 0) t0 = __builtin_readcyclecounter();      // round 1 warms up
    // the layer's weight stream (L2-resident after the first round)
    for (int i = t; i < wstream_vec; i += 512) {
      const f32x4 w = weights[((size_t)b * 7 + (size_t)i) % (size_t)(1 << 16)];
      acc += w;
    }
    if (exchange) {
      store_sc1(mine + t, val + acc);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) {
        store_flag(flags + b * 32, (unsigned)r);
        int spins = 0;
        while (load_flag(flags + partner * 32) < (unsigned)r && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
      const f32x4 got = load_sc1(theirs + t);
      val = got * 0.5f + val * 0.5f;
    } else {
      __syncthreads();
      val = val * 0.999f + acc * 1e-9f;
    }
  }
  if (t == 0) cycles[b] = __builtin_readcyclecounter() - t0;
  if (val[0] == 12345.678f) sink[b] = val[1] + acc[2];
}

int main() {
  const int pairs_list[] = {1, 32, 64};
  f32x4 *box, *weights; unsigned* flags; unsigned long long* cyc; float* sink;
  hipMalloc(&box, 256 * 512 * sizeof(f32x4)); hipMalloc(&weights, (size_t)(1 << 16) * sizeof(f32x4));
  hipMalloc(&flags, 256 * 32 * sizeof(unsigned)); hipMalloc(&cyc, 256 * sizeof(unsigned long long));
  hipMalloc(&sink, 256 * sizeof(float));
  hipMemset(weights, 0, (size_t)(1 << 16) * sizeof(f32x4));
  hipMemset(box, 0, 256 * 512 * sizeof(f32x4));
  const int rounds = 2001;
  int clock_khz = 0;
  hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeWallClockRate, 0);   // s_memrealtime / readcyclecounter base
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("rounds %d, payload 8 KB per direction, 512-thread workgroups, one per CU\n", rounds - 1);
  printf("%-9s %-8s %-10s %-9s %12s %12s\n", "pairs", "xcd", "stream_KB", "exchange", "us/round", "us/exchange");
  for (int pairs : pairs_list)
    for (int same : {1, 0})
      for (int wkb : {0, 288}) {
        double us[2] = {0, 0};
        for (int ex : {0, 1}) {
          // same-XCD pairing needs blocks b and b + 8: launch 16 blocks per pair group so that both exist
          const int blocks = same ? ((2 * pairs + 15) / 16) * 16 : 2 * pairs;
          hipMemset(flags, 0, 256 * 32 * sizeof(unsigned));
          hipDeviceSynchronize();
          hipEventRecord(e0);
          hipLaunchKernelGGL(k_exchange, dim3(blocks), dim3(512), 0, 0, box, flags, weights, rounds, same,
                             wkb * 1024 / 16, cyc, sink, ex);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms = 0; hipEventElapsedTime(&ms, e0, e1);
          us[ex] = 1e3 * ms / rounds;
        }
        printf("%-9d %-8s %-10d %-9s %12.3f %12.3f\n", pairs, same ? "same" : "next", wkb, "yes", us[1], us[1] - us[0]);
      }
  printf("model (config 5, per mc_step): 5 H x H layers + output dot = 6 exchanges; MFMA time saved by halving "
         "each CU's output units = 5 x 128 x 32 cycles / 2.4 GHz = 8.5 us\n");
  return 0;
}
