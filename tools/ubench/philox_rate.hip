// Probe (round 6): shader clocks per Philox4x32-10 call of one wave, with the two 32 x 32 -> 64 bit products of a
// round as v_mul_hi_u32 + v_mul_lo_u32 (common.hpp until round 5) or as one v_mad_u64_u32 each.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/philox_rate.hip -o tools/ubench/philox_rate && tools/ubench/philox_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
template <bool WIDE>
__device__ __forceinline__ void round1(uint4& c, uint2& k) {
  uint32_t hi0, lo0, hi1, lo1;
  if (WIDE) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c.x, p1 = (unsigned long long)0xCD9E8D57u * c.z;
    hi0 = (uint32_t)(p0 >> 32); lo0 = (uint32_t)p0; hi1 = (uint32_t)(p1 >> 32); lo1 = (uint32_t)p1;
  } else {
    hi0 = __umulhi(0xD2511F53u, c.x); lo0 = 0xD2511F53u * c.x; hi1 = __umulhi(0xCD9E8D57u, c.z); lo1 = 0xCD9E8D57u * c.z;
  }
  c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
  k.x += 0x9E3779B9u; k.y += 0xBB67AE85u;
}
template <bool WIDE>
__global__ void k(uint4* out, uint2 key, int calls, unsigned long long* cyc) {
  uint4 c = make_uint4(threadIdx.x, 1, 2, 3);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < calls; ++i) {
    uint2 kk = key;
#pragma unroll
    for (int r = 0; r < 10; ++r) round1<WIDE>(c, kk);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = c;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  uint4* out; unsigned long long* cyc; (void)hipMalloc(&out, 64 * 16); (void)hipMalloc(&cyc, 8);
  uint4 h[2][64];
  for (int w = 0; w < 2; ++w) {
    for (int rep = 0; rep < 2; ++rep) {
      if (w) hipLaunchKernelGGL((k<true>), dim3(1), dim3(64), 0, 0, out, make_uint2(7, 9), 2000, cyc);
      else hipLaunchKernelGGL((k<false>), dim3(1), dim3(64), 0, 0, out, make_uint2(7, 9), 2000, cyc);
      (void)hipDeviceSynchronize();
    }
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h[w], out, 64 * 16, hipMemcpyDeviceToHost);
    printf("%s: %.1f s_memtime ticks per Philox4x32-10 call (one wave)\n", w ? "v_mad_u64_u32       " : "v_mul_hi + v_mul_lo", (double)c / 2000);
  }
  int same = 1;
  for (int i = 0; i < 64; ++i) same &= h[0][i].x == h[1][i].x && h[0][i].y == h[1][i].y && h[0][i].z == h[1][i].z && h[0][i].w == h[1][i].w;
  printf("same bits: %s\n", same ? "yes" : "NO");
  return 0;
}
