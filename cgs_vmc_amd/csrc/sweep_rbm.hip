// k_sweep16 instantiations with the RestrictedBoltzmannNetwork epilogue (log cosh + onsite term).
#include "sweep16.hpp"

hipError_t launch_sweep16_rbm(hipStream_t s, const SweepArgs& a, int Hp) {
#ifdef VMC_QUICK
  return hipErrorInvalidValue;   // development builds leave the RBM sampler out
#else
  return launch_sweep16_r<true>(s, a, Hp);
#endif
}
