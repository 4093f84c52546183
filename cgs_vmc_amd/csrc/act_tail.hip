// Row kernels (k_tail16 / k_tail0 / k_tail_lds / k_backprop16) instantiated for ONE hidden activation:
// compiled once per id of layers.NONLINEARITIES with -DVMC_INST_ACT=<id> (csrc/Makefile).
#include "tail16.hpp"
#include "tail_lds.hpp"

#ifndef VMC_INST_ACT
#error "compile with -DVMC_INST_ACT=<activation id>"
#endif
#define VMC_CAT2(a, b) a##b
#define VMC_CAT(a, b) VMC_CAT2(a, b)

hipError_t VMC_CAT(launch_tail_inst_, VMC_INST_ACT)(hipStream_t s, const TailArgs& a, int Hp,
                                                    bool ratio_mode, bool rbm) {
  return launch_tail_act<VMC_INST_ACT>(s, a, Hp, ratio_mode, rbm);
}

hipError_t VMC_CAT(launch_backprop16_inst_, VMC_INST_ACT)(hipStream_t s, const float* act_all,
                                                          float* delta_all, const float* p16t,
                                                          const float* woutp, int B, int Hp,
                                                          int n_hidden, bool rbm,
                                                          const float* dact_all,
                                                          const float* oscale, const ElocFold& ef,
                                                          const OutLayerSums& out) {
  return launch_backprop16_t<VMC_INST_ACT>(s, act_all, delta_all, p16t, woutp, B, Hp, n_hidden, rbm,
                                           dact_all, oscale, ef, out);
}

// 257 .. 512 hidden units (384 / 512 padded): the LDS-operand row kernel of tail_lds.hpp
hipError_t VMC_CAT(launch_tail_lds_inst_, VMC_INST_ACT)(hipStream_t s, const TailArgs& a, int Hp,
                                                        bool ratio_mode, bool rbm) {
  return launch_tail_lds_act<VMC_INST_ACT>(s, a, Hp, ratio_mode, rbm);
}
