// k_sweep16 instantiations of the fully-connected ansatz + the public launcher.
#include "sweep16.hpp"

// LDS the sampler needs at least (W1 streamed from L2); vmc_create rejects shapes beyond 160 KiB
size_t sweep_lds_required(int N, int Hp, int n_hidden, bool rbm) {
  return sweep_lds_bytes(N, Hp, n_hidden, false, rbm);
}


hipError_t launch_sweep16_rbm(hipStream_t s, const SweepArgs& a, int Hp);   // sweep_rbm.hip

hipError_t launch_sweep16(hipStream_t s, const SweepArgs& a, int Hp) {
  if (a.B <= 0) return hipSuccess;
  if (a.rbm) return launch_sweep16_rbm(s, a, Hp);
  return launch_sweep16_r<false>(s, a, Hp);
}
