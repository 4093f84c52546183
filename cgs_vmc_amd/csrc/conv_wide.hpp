// Convolutional ansatz kernels for more than 16 filters: the templates of conv_kernels.hpp with
// CONV_NCB = 2, 3 or 4 channel blocks of 16 (layers.py:89-160 takes any num_conv_filters, utils.py:111),
// one translation unit per block count (conv32.hip, conv48.hip, conv64.hip define CONV_NCB and include
// this file) so that they compile side by side.
#ifndef CONV_NCB
#error "define CONV_NCB (2, 3 or 4) before including conv_wide.hpp"
#endif
#define CONV_WAVES 8     // one 8-wave workgroup per CU (see conv_kernels.hpp)
#include "conv_kernels.hpp"

#define CONV_CAT_(a, b) a##b
#define CONV_CAT(a, b) CONV_CAT_(a, b)
#define CONV_CB(name) CONV_CAT(CONV_CAT(conv_launch_##name, _cb), CONV_NCB)

hipError_t CONV_CB(rows)(hipStream_t s, const ConvRowsArgs& a, dim3 grid, size_t lds) {
  return conv_launch_rows_t<CONV_NCB>(s, a, grid, lds);
}
hipError_t CONV_CB(sweep)(hipStream_t s, const ConvSweepArgs& a, dim3 grid, size_t lds) {
  return conv_launch_sweep_t<CONV_NCB>(s, a, grid, lds);
}
hipError_t CONV_CB(back)(hipStream_t s, const ConvBackArgs& a, dim3 grid, size_t lds) {
  return conv_launch_back_t<CONV_NCB>(s, a, grid, lds);
}
hipError_t CONV_CB(dw)(hipStream_t s, const ConvDwArgs& a, dim3 grid, size_t lds) {
  return conv_launch_dw_t<CONV_NCB>(s, a, grid, lds);
}
hipError_t CONV_CB(sr_rowdot)(hipStream_t s, const ConvSrRowdotArgs& a, dim3 grid, size_t lds) {
  return conv_launch_sr_rowdot_t<CONV_NCB>(s, a, grid, lds);
}
