// k_sweep16 instantiated for ONE hidden activation and ONE output epilogue:
//   -DVMC_INST_ACT=<id> -DVMC_INST_RBM=0   fully-connected epilogue + the public per-activation launcher
//   -DVMC_INST_ACT=<id> -DVMC_INST_RBM=1   RestrictedBoltzmannNetwork epilogue (log cosh + onsite)
#include "sweep16.hpp"

#if !defined(VMC_INST_ACT) || !defined(VMC_INST_RBM)
#error "compile with -DVMC_INST_ACT=<activation id> -DVMC_INST_RBM=<0|1>"
#endif
#define VMC_CAT2(a, b) a##b
#define VMC_CAT(a, b) VMC_CAT2(a, b)

#if VMC_INST_RBM
hipError_t VMC_CAT(launch_sweep16_rbm_inst_, VMC_INST_ACT)(hipStream_t s, const SweepArgs& a, int Hp) {
#ifdef VMC_QUICK
  return hipErrorInvalidValue;   // development builds leave the RBM sampler out
#else
  return launch_sweep16_r<true, VMC_INST_ACT>(s, a, Hp);
#endif
}
#else
hipError_t VMC_CAT(launch_sweep16_rbm_inst_, VMC_INST_ACT)(hipStream_t s, const SweepArgs& a, int Hp);

hipError_t VMC_CAT(launch_sweep16_inst_, VMC_INST_ACT)(hipStream_t s, const SweepArgs& a, int Hp) {
  if (a.rbm) return VMC_CAT(launch_sweep16_rbm_inst_, VMC_INST_ACT)(s, a, Hp);
  return launch_sweep16_r<false, VMC_INST_ACT>(s, a, Hp);
}
#endif
