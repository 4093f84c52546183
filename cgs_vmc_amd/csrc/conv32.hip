// Convolutional ansatz kernels for 17 .. 32 filters: the templates of conv_kernels.hpp with NCB = 2
// channel blocks (layers.py:89-160 takes any num_conv_filters, utils.py:111).
#define CONV_WAVES 8     // one 8-wave workgroup per CU (see conv_kernels.hpp)
#include "conv_kernels.hpp"

hipError_t conv_launch_rows_cb2(hipStream_t s, const ConvRowsArgs& a, dim3 grid, size_t lds) {
  return conv_launch_rows_t<2>(s, a, grid, lds);
}
hipError_t conv_launch_sweep_cb2(hipStream_t s, const ConvSweepArgs& a, dim3 grid, size_t lds) {
  return conv_launch_sweep_t<2>(s, a, grid, lds);
}
hipError_t conv_launch_back_cb2(hipStream_t s, const ConvBackArgs& a, dim3 grid, size_t lds) {
  return conv_launch_back_t<2>(s, a, grid, lds);
}
hipError_t conv_launch_dw_cb2(hipStream_t s, const ConvDwArgs& a, dim3 grid, size_t lds) {
  return conv_launch_dw_t<2>(s, a, grid, lds);
}
hipError_t conv_launch_sr_rowdot_cb2(hipStream_t s, const ConvSrRowdotArgs& a, dim3 grid, size_t lds) {
  return conv_launch_sr_rowdot_t<2>(s, a, grid, lds);
}
