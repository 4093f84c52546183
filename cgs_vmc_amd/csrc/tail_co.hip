// k_tail_co: the local-energy row kernel that shares a CU with the sampler (fully_connected, relu,
// H = 256).  operators.py:137-169 / 249-259: one row per connected configuration, value
// 0.5 jx psi(row) / psi(chain).
//
// Status: measured, NOT the default (CGS_VMC_CO=1 selects it; DESIGN.md 5 has the numbers): the pair
// is resident together as designed, but the hardware arbitrates the matrix pipe per instruction, not
// per phase -- this kernel's MFMAs also delay the sampler's own MFMA phases, which are its critical
// path (s_setprio makes no difference), and the step comes out at 2.07 ms against 2.04 ms.
//
// Why it exists.  At config 3 (4096 chains) the sampler k_sweep16 has exactly one 16-chain tile per
// CU and its mc_step is one dependent chain, so the matrix pipe idles ~35 % of the sweep; k_tail16
// (402 registers, one wave per SIMD) cannot be resident next to it.  This kernel computes the same
// rows in 112 registers and half the LDS: with k_sweep16_co (2 x 200 registers per SIMD, W1 in L2)
// one workgroup of each fits a CU, and the hardware issues this kernel's MFMAs whenever the
// sampler's waves wait (proposals, z1' build, output dot, accept, barriers).
//
// Shape: 4 waves, 32 rows per tile (two 16-row halves that share every weight fragment).  Wave w
// owns output units 64 w .. 64 w + 63 (TO = 4 tiles) of both halves: 8 accumulator tiles; the B
// operands of a layer come from LDS (as in the sampler), the A fragments stream from L2 through a
// two-stage register ring (one k-tile = 4 KiB per wave = 32 MFMAs ahead).  One barrier per layer.
//
// The same body (tail_lds.hpp), without the register cap and for 24 or 32 unit tiles per row (384 /
// 512 hidden units: TO = 6 / 8 tiles per wave), is the row kernel of the fused path for 257 .. 512
// hidden units (k_tail_lds, instantiated per activation in act_tail.hip).
#include "tail_lds.hpp"

#ifndef TAILCO_VGPR
#define TAILCO_VGPR 56   // amdgpu_num_vgpr counts half of the unified file: 112 registers
#endif

__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(TAILCO_VGPR)))
void k_tail_co(TailArgs a) { tail_lds::tail_lds_body<16, true, false, VMC_ACT_RELU_>(a); }

size_t tail_co_lds_bytes(int n_hidden) { return tail_lds::lds_bytes(16, n_hidden); }

// lds_bytes >= tail_co_lds_bytes(n_hidden): the caller pads it beyond half a CU's LDS so that two
// of these workgroups never share a CU (the second one would take the sampler's place)
hipError_t launch_tail_co(hipStream_t s, const TailArgs& a, size_t lds_bytes) {
  if (a.n_rows <= 0) return hipSuccess;
  if (a.n_hidden < 1 || lds_bytes < tail_co_lds_bytes(a.n_hidden) || lds_bytes > 160 * 1024)
    return hipErrorInvalidValue;
  const int tiles = (a.n_rows + 31) / 32;
  const int persistent = a.num_cus > 0 ? a.num_cus : 256;
  const dim3 grid(tiles < persistent ? tiles : persistent), block(256);
  hipError_t e = hipFuncSetAttribute((const void*)k_tail_co,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_tail_co, grid, block, lds_bytes, s, a);
  return hipGetLastError();
}

// 384 or 512 padded units and at least one H x H layer (the sampler's LDS need is checked by vmc_create)
bool tail_lds_supported(int Hp, int n_hidden) {
  return (Hp == 384 || Hp == 512) && n_hidden >= 1 && tail_lds::lds_bytes(Hp / 16, n_hidden) <= 160 * 1024;
}
