// k_sweep16s -- EXPERIMENT (CGS_VMC_SPLIT_BF16=2): the persistent sampler of sweep16.hpp with its
// H x H layers as 3 x bf16 split products on the BF16 matrix cores (template flag SW there).  256 relu units,
// fully_connected, lattices of at most 256 sites; injected steps and the proposal dump take the general variant.
#include "sweep16.hpp"

#ifndef SWEEP_SPLIT_RT
#define SWEEP_SPLIT_RT 8     // resident 16-unit k-tiles of layer 0 = 4 resident 32-deep k-steps (96 registers)
#endif

bool sweep16_split_supported(int N, int Hp, int n_hidden) {
  return Hp == 256 && n_hidden >= 1 && (N + 3) / 4 <= 64 &&
         plan_sweep_lds_bytes(N, Hp, n_hidden, false, false, 0, true) <= PLAN_LDS_PER_CU;
}

hipError_t launch_sweep16_split(hipStream_t s, const SweepArgs& a_in) {
  SweepArgs a = a_in;
  if (a.B <= 0) return hipSuccess;
  if (!a.p16s || a.rbm || a.act != VMC_ACT_RELU_ || !sweep16_split_supported(a.N, 256, a.n_hidden)) return hipErrorInvalidValue;
  const dim3 grid((a.B + 15) / 16), block(512);
  const size_t lds = plan_sweep_lds_bytes(a.N, 256, a.n_hidden, false, false, 0, true);
  a.uh_lds = 0;
  const bool plain = a.inj_up == nullptr && a.dbg_up == nullptr;
  hipError_t e;
  if (plain && (a.N + 3) / 4 > 32) {       // 129 .. 256 sites: four prefetched site blocks per lane (config 5)
    e = hipFuncSetAttribute((const void*)k_sweep16s<SWEEP_SPLIT_RT, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_sweep16s<SWEEP_SPLIT_RT, true, 4>), grid, block, lds, s, a);
  } else if (plain) {
    e = hipFuncSetAttribute((const void*)k_sweep16s<SWEEP_SPLIT_RT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_sweep16s<SWEEP_SPLIT_RT, true>), grid, block, lds, s, a);
  } else {
    e = hipFuncSetAttribute((const void*)k_sweep16s<SWEEP_SPLIT_RT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_sweep16s<SWEEP_SPLIT_RT, false>), grid, block, lds, s, a);
  }
  return hipGetLastError();
}
