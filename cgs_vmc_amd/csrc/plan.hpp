// plan.hpp -- every PURE-HOST decision of libcgsvmc_hip.so in one header with no HIP in it: parameter
// layouts, vmc_create's shape / LDS validation, the convolution group / band / slice pickers, the
// sampler's LDS plan and kernel-variant choice, split-K and the XCD-aware block order of the
// weight-gradient GEMM, the stochastic-reconfiguration tile schedules, and the sizes of the buffers
// they index.  The .hip files call these functions (there is no second copy of any formula); the same
// header is compiled by g++ -fsanitize=address,undefined into hostcheck.cpp, which walks a grid of
// shapes on the CPU and asserts that every offset, grid and LDS figure is in range
// (`make -C cgs_vmc_amd/csrc hostcheck`, tests/test_hostcheck.py; SURVEY.md 5 "sanitizers").
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/cgsvmc.h"

#if defined(__HIPCC__)
#define PLAN_HD __host__ __device__
#else
#define PLAN_HD
#endif

#define PLAN_LDS_PER_CU ((size_t)160 * 1024)   // gfx950: 160 KiB of LDS per CU

// ------------------------------------------------------------------------------- dense parameter layout
// offsets of the pieces of the flat parameter vector (both dense ansatz types; include/cgsvmc.h)
struct ParamLayout {
  long long off_w1, off_b1;   // first layer [N,H], [H]
  long long off_h0;           // first H x H layer; layer l at off_h0 + l (H*H + H): w then b
  long long off_wout, off_bout;  // FC: w_out [H], b_out;  RBM: off_wout = -1, off_bout = b_on
  long long off_won;          // RBM: onsite weights [N]; FC: -1
  int n_hh;                   // number of H x H layers (FC: L-1, RBM: L)
};

inline ParamLayout plan_layout(bool rbm, long long N, long long H, long long L) {
  ParamLayout lay;
  if (rbm) {   // w_on[N] b_on | w_1 b_1 | (w b) x L        (see include/cgsvmc.h)
    lay.off_won = 0; lay.off_bout = N; lay.off_wout = -1;
    lay.off_w1 = N + 1; lay.off_b1 = lay.off_w1 + N * H; lay.off_h0 = lay.off_b1 + H;
    lay.n_hh = (int)L;
  } else {     // w_1 b_1 | (w b) x (L-1) | w_out b_out
    lay.off_w1 = 0; lay.off_b1 = N * H; lay.off_h0 = lay.off_b1 + H;
    lay.n_hh = (int)L - 1;
    lay.off_wout = lay.off_h0 + (L - 1) * (H * H + H); lay.off_bout = lay.off_wout + H;
    lay.off_won = -1;
  }
  return lay;
}

// weight matrix / bias of layer l (0 = first layer, 1 .. n_hh = the H x H layers)
inline long long plan_off_w(const ParamLayout& lay, long long H, int l) {
  return l == 0 ? lay.off_w1 : lay.off_h0 + (long long)(l - 1) * (H * H + H);
}
inline long long plan_off_b(const ParamLayout& lay, long long H, int l) {
  return l == 0 ? lay.off_b1 : plan_off_w(lay, H, l) + H * H;
}

inline long long plan_num_params_dense(int ansatz, long long N, long long H, long long L) {
  if (ansatz == VMC_ANSATZ_RBM) return N + 1 + N * H + H + L * (H * H + H);
  return N * H + H + (L - 1) * (H * H + H) + H + 1;
}

inline long long plan_num_params_conv(int n_conv, long long F, long long taps) {
  return taps * F + F + (long long)(n_conv - 1) * (taps * F * F + F);
}

// ------------------------------------------------------------------------------- convolutional geometry
#define CONV_MAX_LAYERS 32
#define CONV_FP 16          // channel tile of the MFMA: filters are zero padded to NCB blocks of 16
#define CONV_MAX_NCB 4      // num_conv_filters <= 64
#define CONV_GENERAL_MAX_K 31   // general path (conv_general.hip): any kernel size the periodic padding allows, bounded for the index arithmetic
#define CONV_GENERAL_MAX_F 1024
#define CONV_MAX_K 9        // kernel_size (one instantiation per size; weights of a block pair in register-resident chunks of <= 25 taps)
#define CONV_LDS_PER_WG ((size_t)80 * 1024)   // two 4-wave workgroups share the 160 KiB of a CU

// Geometry of one network.  A feature map of one sample is stored -- in LDS and in the HBM tapes
// alike -- channel-group major: [4 NCB groups][GS dwords], element (site, channel c) at
// (c / 4) * GS + 4 * site + c % 4, with GS >= 4 N a multiple of 64 dwords so that the 16-lane
// groups of ds_read_b128 / ds_write_b128 fall on distinct banks; NCB = ceil(F / 16) channel blocks
// of one MFMA tile each, CS = 4 NCB GS dwords per sample.
struct ConvGeom {
  int K;        // kernel_size
  int D1, D2;   // size_x, size_y: inputs are reshaped to [-1, size_x, size_y, 1] (wavefunctions.py:596)
  int N;        // D1 * D2
  int F;        // num_conv_filters
  int n_conv;   // number of Conv2dPeriodic modules: num_conv_layers, or 1 + 2 num_resnet_blocks
  int resnet;   // 0: Conv2DNetwork, 1: ResNet2D
  int hact;     // hidden activation id of Conv2DNetwork (ResNet2D: selu, layers.py:226)
  int GS;       // dwords per channel group of a feature map (see above)
  int lo, hi;   // periodic padding in front / behind along axis 1: (K-1)/2 and K/2 (layers.py:132-141);
                // the 1-D modules pad K/2 in front and K-1-K/2 behind (layers.py:66-72)
  int KW;       // taps along axis 2: K (Conv2dPeriodic) or 1 (Conv1dPeriodic on an [N, 1] lattice)
  int lo2, hi2; // the same padding for axis 2 (0 for the 1-D modules)
  int NCB;      // channel blocks of 16: (F + 15) / 16
  int CS;       // dwords of one sample's feature map: 4 * NCB * GS
};

// LDS of a row / sampler / backward workgroup holding G samples: buf0, buf1, xs, pinfo, row_chain,
// red + the sampler's cur_logit, prop, prop_u + the two wrap tables
// ints per row of the periodic neighbour tables (one entry per tap along an axis)
// general path: row stride (floats) of the im2col matrix = taps x input channels of the widest convolution, 16-byte rows
PLAN_HD inline int plan_cgen_lda(const ConvGeom& g) { return (g.K * g.KW * (g.n_conv > 1 ? g.F : 1) + 3) & ~3; }
PLAN_HD inline int plan_conv_tab(const ConvGeom& g) { return g.K <= 8 ? 8 : 16; }
// general path at <= 16 filters: k_cgen_band (conv_band.hip) stages bands of lattice rows with their periodic halo,
// (rows + K - 1) x (D2 + KW - 1) sites x 16 channels, in LDS; rows per band = as many as keep a band within
// PLAN_CGEN_BAND_LDS bytes (several workgroups per CU); < 1: the lattice is too wide for a band
#define PLAN_CGEN_BAND_LDS (40 * 1024)
#define PLAN_CGEN_BAND_LDS_WIDE (144 * 1024)
#define PLAN_CGEN_BAND_MAX_FRAGS 208      // weight fragments (registers) of one output block: K KW 4 NCB
inline int plan_cgen_band_ncb(const ConvGeom& g) { return (g.F + 15) / 16; }
inline int plan_cgen_band_rows(const ConvGeom& g) {
  const long long per_row = (long long)(g.D2 + g.KW - 1) * 16 * plan_cgen_band_ncb(g) * (long long)sizeof(float);
  // one block: 40 KB (several workgroups per CU); more: one workgroup per CU stages a band for all of its output blocks, up to 144 KB
  long long bh = (plan_cgen_band_ncb(g) > 1 ? PLAN_CGEN_BAND_LDS_WIDE : PLAN_CGEN_BAND_LDS) / per_row - (g.K - 1);
  if (bh > g.D1) bh = g.D1;
  return (int)bh;
}
// rows per band of ONE launch over `rows` row configurations on a chip that holds `capacity` workgroups: few rows (the
// sampler's B candidates at a small batch) take thinner bands, so that every CU has an item -- a launch's latency is an
// item's; many rows (the local energies) take the widest band (least halo re-read)
inline int plan_cgen_band_rows_for(const ConvGeom& g, long long rows, long long capacity) {
  const int bh_max = plan_cgen_band_rows(g);
  if (bh_max < 1 || rows < 1) return bh_max;
  const long long nb_max = (g.D1 + bh_max - 1) / bh_max;
  if (rows * nb_max >= capacity) return bh_max;
  long long nb = (capacity + rows - 1) / rows;
  if (nb > g.D1) nb = g.D1;
  long long bh = (g.D1 + nb - 1) / nb;
  if (bh < 1) bh = 1;
  return (int)(bh < bh_max ? bh : bh_max);
}
// the shapes k_cgen_band takes: up to 64 filters (four channel blocks of 16) as long as one output block's fragments
// against every input block fit the registers (3 x 3: 64 filters; 4 x 4: 48; 5 x 5: 32; 7 x 7: 16), 2 .. 7 taps per
// axis (2-D: K x K; 1-D: K x 1), a band of at least one lattice row within the LDS budget
inline bool plan_cgen_band_ok(const ConvGeom& g) {
  if (g.F < 1 || g.F > 64 || g.K < 2 || g.K > 7) return false;
  if (!(g.KW == g.K || g.KW == 1)) return false;
  if (g.K * g.KW * 4 * plan_cgen_band_ncb(g) > PLAN_CGEN_BAND_MAX_FRAGS) return false;
  return plan_cgen_band_rows(g) >= 1;
}
// Chain groups of the general convolution sampler (run_sweep_cgen: a step of each group on a stream of its own).  Two
// groups where a launch over the whole batch leaves CUs idle that the next launch of the other group can use: the GEMM
// form with more than one round of 128-position row tiles whose last round is less than 0.9 full (1,024 chains on a
// 10 x 10 lattice: 800 tiles on 256 CUs, 3.125 rounds paid as 4), and the one-workgroup-per-CU band kernel from one
// round of positions on (its tail: measured, 36 x 36 x 64 filters at 32 chains: 161 -> 151 ms per sweep).  One group where
// the launches are latency (36 x 36 x 16 filters at 32 chains: 68 -> 73 ms with two, 214 with four -- the host's launch rate).
inline int plan_cgen_sweep_groups(const ConvGeom& g, long long B, int num_cus) {
  if (B < 2 || num_cus < 1) return 1;
  const long long positions = B * g.N;
  if (plan_cgen_band_ok(g)) return (plan_cgen_band_ncb(g) > 1 && positions >= 128LL * num_cus) ? 2 : 1;
  const long long tiles = ((positions + 127) / 128) * ((g.F + 127) / 128);
  if (tiles <= num_cus) return 1;
  const long long rounds = (tiles + num_cus - 1) / num_cus;
  return tiles * 10 < rounds * num_cus * 9 ? 2 : 1;
}
// The patch sampler of the general convolution path (k_cgen_patch_sweep, conv_patch.hip): an exchange negates two spins,
// and convolution l's output changes only in a box of (l + 1)(K - 1) + 1 sites per axis around each of them (the union of
// the taps' reach: graph_builders.py:67-71 proposes, layers.py:118-160 convolves).  One workgroup per chain keeps the
// chain's maps of every convolution in HBM and recomputes the two boxes per convolution instead of the lattice.
// Shapes: Conv2DNetwork / Conv1DNetwork / ResNet2D / ResNet1D, the band kernel's single-block shapes (<= 16 filters, so that a box
// value has k_cgen_band's bits), at least two convolutions, the last box within the lattice along both axes (it must not
// meet itself around the torus), and the LDS within a CU's.
inline int plan_cgen_patch_side(const ConvGeom& g, int l, int axis) { return (l + 1) * ((axis ? g.KW : g.K) - 1) + 1; }
inline size_t plan_cgen_patch_lds_bytes(const ConvGeom& g) {
  const int L = g.n_conv, T = g.K * g.KW;
  size_t fl = (size_t)((g.N + 3) & ~3);                           // the chain's spins
  fl += (size_t)(L - 1) * T * 256;                                 // weight fragments of the convolutions behind the first
  fl += (size_t)L * 16;                                            // biases
  size_t win = (size_t)(plan_cgen_patch_side(g, 0, 0) + g.K - 1) * (size_t)(plan_cgen_patch_side(g, 0, 1) + g.KW - 1);   // first: 1 channel
  for (int l = 0; l < L; ++l) {
    fl += 2 * (size_t)plan_cgen_patch_side(g, l, 0) * plan_cgen_patch_side(g, l, 1) * 16;          // the two boxes of convolution l
    if (l > 0) {
      const size_t w = (size_t)(plan_cgen_patch_side(g, l, 0) + g.K - 1) * (size_t)(plan_cgen_patch_side(g, l, 1) + g.KW - 1) * 16;
      if (w > win) win = w;
    }
  }
  fl += 2 * win;                                                   // the two staged input windows
  fl += (size_t)((g.N + 7) / 8 * 4);                               // 16-bit marks of the sites inside the last convolution's boxes
  fl += (size_t)((g.N + 3) & ~3);                                  // the next step's site uniforms
  return fl * sizeof(float) + 256;                                 // + the step's scalars
}
#define PLAN_CGEN_PATCH_LDS (156 * 1024)
inline bool plan_cgen_patch_ok(const ConvGeom& g, long long B) {
  if (g.n_conv < 2 || g.n_conv > 9 || g.F > 16 || !plan_cgen_band_ok(g)) return false;       // (residual networks: 1 + 2 blocks convolutions)
  if (plan_cgen_patch_side(g, g.n_conv - 1, 0) > g.D1 || plan_cgen_patch_side(g, g.n_conv - 1, 1) > g.D2) return false;
  if (g.N > 16384 || B < 1) return false;
  if ((long long)g.n_conv * B * g.N * ((g.F + 3) & ~3) * (long long)sizeof(float) > (4LL << 30)) return false;   // the chains' maps
  return plan_cgen_patch_lds_bytes(g) <= PLAN_CGEN_PATCH_LDS;
}
// ... and where it pays: the boxes of the last convolution cover at most half of the lattice
inline bool plan_cgen_patch_pays(const ConvGeom& g) {
  return 4LL * plan_cgen_patch_side(g, g.n_conv - 1, 0) * plan_cgen_patch_side(g, g.n_conv - 1, 1) <= g.N;
}
// ... and where they beat the FUSED kernels (whose maps never leave the LDS): the boxes of all convolutions together at most
// a fifth of the positions a forward computes.  Whole steps at 256 chains, fused -> general path with the patch kernels:
// 24 x 24, 2 x 16 filters 5 x 5 (18 %): 36.2 -> 19.8 ms; 20 x 20, 3 x 16 filters 3 x 3 (14 %): 17.6 -> 13.6 ms.  plan_desc sends
// such a shape to the general path although the fused kernels would take it (CGS_VMC_CONV_GENERAL=0: not by preference).
inline bool plan_cgen_patch_routes(const ConvGeom& g, long long B) {
  if (g.resnet || !plan_cgen_patch_ok(g, B) || !plan_cgen_patch_pays(g)) return false;      // (residual networks: not measured against their fused kernels)
  long long boxes = 0;
  for (int l = 0; l < g.n_conv; ++l) boxes += 2LL * plan_cgen_patch_side(g, l, 0) * plan_cgen_patch_side(g, l, 1);
  return 5 * boxes <= (long long)g.n_conv * g.N;
}
// k_cgen_first_direct (conv_band.hip): spins [N], weights [taps][Fp], bias [Fp], neighbour table [N][taps]
inline size_t plan_cgen_first_direct_lds_bytes(const ConvGeom& g) {
  const size_t fp = (size_t)((g.F + 3) & ~3), t = (size_t)g.K * g.KW;
  return sizeof(float) * ((size_t)((g.N + 3) & ~3) + t * fp + fp + (size_t)g.N * t);
}
inline size_t plan_cgen_band_lds_bytes(const ConvGeom& g, bool first, int band_rows = 0) {
  const int bh = band_rows > 0 ? band_rows : plan_cgen_band_rows(g);
  return (size_t)(bh + g.K - 1) * (size_t)(g.D2 + g.KW - 1) * (first ? 1 : 16 * (size_t)plan_cgen_band_ncb(g)) * sizeof(float);
}
inline size_t plan_conv_rows_lds(const ConvGeom& g, int G) {
  const size_t xs = (size_t)((g.N + 3) & ~3);
  return ((size_t)G * 2 * g.CS + (size_t)G * xs + (size_t)G * g.N + (size_t)G * 7 + 2 * (size_t)plan_conv_tab(g) * (size_t)(g.D1 + g.D2) + 16) * sizeof(float);
}

// LDS budget of one workgroup: half a CU when a sample's feature maps allow two workgroups per CU
inline size_t plan_conv_lds_cap(const ConvGeom& g, bool one_wg_per_cu = false) {
  if (one_wg_per_cu || g.NCB > 1) return PLAN_LDS_PER_CU;      // one 8-wave workgroup per CU (conv_wide.hpp)
  return plan_conv_rows_lds(g, 1) <= CONV_LDS_PER_WG ? CONV_LDS_PER_WG : PLAN_LDS_PER_CU;
}

inline int plan_conv_waves(const ConvGeom& g) { return g.NCB > 1 ? 8 : 4; }

// samples per pass of the row / backward kernels: the group size (<= 64, LDS within the cap) whose
// position tiles divide most evenly over the waves; ties go to the larger group
inline int plan_conv_pick_group(const ConvGeom& g, int waves, bool one_wg_per_cu = false) {
  int best = 1; double best_eff = -1.0;
  for (int G = 1; G <= 64; ++G) {
    if (plan_conv_rows_lds(g, G) > plan_conv_lds_cap(g, one_wg_per_cu)) break;
    const int tiles = (G * g.N + 15) / 16;
    const int rounds = (tiles + waves - 1) / waves;
    const double eff = (double)G * g.N / 16.0 / ((double)rounds * waves);
    if (eff >= best_eff - 1e-9) { best = G; best_eff = eff; }
  }
  return best;
}

// sampler: chains per workgroup that minimise (workgroups per CU) x (tile rounds of one forward pass) --
// the MFMA time of the busiest CU per mc_step.  The two-channel-block kernels walk their tiles in
// pairs.  Ties go to the group size that fills the last round of workgroups best (4096 chains, 32
// filters on 10 x 10: G = 4 -> 1024 workgroups = 4 per CU measured 84.9 ms per sweep, G = 5 -> 820
// workgroups 87.8 ms), then to the larger group.
inline int plan_conv_pick_sweep_group(const ConvGeom& g, long long B, int num_cus, int waves,
                                      bool one_wg_per_cu = false) {
  long long best_cost = -1;
  double best_fill = 0.0;
  int best = 1;
  for (int G = 1; G <= 64 && plan_conv_rows_lds(g, G) <= plan_conv_lds_cap(g, one_wg_per_cu) && G <= B; ++G) {
    const long long wgs = (B + G - 1) / G, per_cu = (wgs + num_cus - 1) / num_cus;
    long long tiles = ((long long)G * g.N + 15) / 16;
    if (g.NCB > 1) tiles = (tiles + 1) / 2;          // tile pairs
    const long long tile_rounds = (tiles + waves - 1) / waves;
    const long long cost = per_cu * tile_rounds;
    const double fill = (double)wgs / (double)(per_cu * num_cus);
    if (best_cost < 0 || cost < best_cost || (cost == best_cost && fill >= best_fill - 1e-9)) {
      best_cost = cost; best_fill = fill; best = G;
    }
  }
  return best;
}

// persistent grids of the row / backward / SR row-dot kernels: one workgroup per resident slot
inline int plan_conv_grid(const ConvGeom& g, long long n_rows, int G, int num_cus) {
  const long long groups = (n_rows + G - 1) / G;
  const long long slots = (long long)num_cus * ((g.NCB == 1 && plan_conv_rows_lds(g, G) <= CONV_LDS_PER_WG) ? 2 : 1);
  return (int)(groups < slots ? groups : slots);
}

// LDS of the weight-gradient kernel for bands of `rows` lattice rows: delta [NQ][CW] + input
// [NIN][CW] in one padded site numbering, the halo and position maps, ones (+ 8 sites: the operands
// of the quad past the end are read, and dropped)
// k_conv_dw keeps (items per wave) x (input blocks) x (output blocks) accumulator pairs in registers
// (items of a layer: the taps and the bias, over PLAN_DW_WAVES waves): at most PLAN_DW_MAX_ACC of them.
// A workgroup therefore takes NCO of the NCB output channel blocks (all of them when that fits with at most
// two blocks, else one) and one of NTP parts of the items; grid z = (NCB / NCO) * NTP.
#define PLAN_DW_WAVES 8
#define PLAN_DW_MAX_ACC 24
PLAN_HD constexpr int plan_conv_dw_nco(int K, int KW, int NCB) {
  return (NCB <= 2 && ((K * KW + 1 + PLAN_DW_WAVES - 1) / PLAN_DW_WAVES) * NCB * NCB <= PLAN_DW_MAX_ACC) ? NCB : 1;
}
PLAN_HD constexpr int plan_conv_dw_parts(int K, int KW, int NCB) {
  const int nco = plan_conv_dw_nco(K, KW, NCB), ni = K * KW + 1;
  int p = 1;
  while (((ni + PLAN_DW_WAVES * p - 1) / (PLAN_DW_WAVES * p)) * NCB * nco > PLAN_DW_MAX_ACC) ++p;
  return p;
}
inline int plan_conv_dw_nco(const ConvGeom& g) { return plan_conv_dw_nco(g.K, g.KW, g.NCB); }
inline int plan_conv_dw_grid_z(const ConvGeom& g) {
  return (g.NCB / plan_conv_dw_nco(g)) * plan_conv_dw_parts(g.K, g.KW, g.NCB);
}
inline size_t plan_conv_dw_lds(const ConvGeom& g, int rows) {
  const size_t d2p = (size_t)g.D2 + g.KW - 1, npad = (size_t)(g.D1 + g.K - 1) * d2p;
  const size_t nq = ((size_t)rows * d2p + 3) & ~(size_t)3, nin = nq + (size_t)(g.K - 1) * d2p + g.KW;
  const size_t cw = 16 * (size_t)g.NCB, cwd = 16 * (size_t)plan_conv_dw_nco(g);
  return (nq * cwd + nin * cw + npad + g.N + cw + 8 * cw) * sizeof(float);
}

// rows per band: the whole sample when it fits the CU's LDS, else the largest band that does
// (`forced` >= 1: the test knob CGS_VMC_CONV_DW_BAND); 0 when not even one row fits
inline int plan_conv_dw_band(const ConvGeom& g, int forced = 0) {
  int rows = g.D1;
  if (forced >= 1 && forced < rows) rows = forced;
  while (rows > 1 && plan_conv_dw_lds(g, rows) > PLAN_LDS_PER_CU) --rows;
  return plan_conv_dw_lds(g, rows) <= PLAN_LDS_PER_CU ? rows : 0;
}

// sample slices of the weight-gradient kernel: two resident workgroups per CU (NCB = 1) over the
// layers > 0, whose workgroups carry the work
inline int plan_conv_dw_slices(const ConvGeom& g, long long B, int num_cus) {
  const int per_cu = g.NCB == 1 ? 2 : 1;
  const int heavy = (g.n_conv > 1 ? g.n_conv - 1 : 1) * plan_conv_dw_grid_z(g);
  int sl = (per_cu * num_cus + heavy - 1) / heavy;
  sl = sl < 64 ? 64 : (sl > 256 ? 256 : sl);
  return B < sl ? (int)B : sl;
}

// floats of the weight-gradient workspace [slices][n_conv][2][(taps * 16 NCB + 1) * 16 NCB]
inline long long plan_conv_dw_ws_floats(const ConvGeom& g, int slices) {
  const long long KK = (long long)g.K * g.KW, cw = 16LL * g.NCB;
  return (long long)slices * g.n_conv * 2 * (KK * cw + 1) * cw;
}

// floats of the packed images of one parameter set: w0, wf / wb (each), bias
inline long long plan_conv_w0_floats(const ConvGeom& g) { return (long long)g.NCB * (((long long)g.K * g.KW + 3) / 4) * 64; }
inline long long plan_conv_wf_floats(const ConvGeom& g) {
  const long long nl = g.n_conv > 1 ? g.n_conv - 1 : 1;
  return nl * g.NCB * g.NCB * (long long)g.K * g.KW * 256;
}
inline long long plan_conv_bias_floats(const ConvGeom& g) { return (long long)g.n_conv * 16 * g.NCB; }

// ------------------------------------------------------------------------------- fused dense kernels
// LDS of the sampler k_sweep16 (16 chains per workgroup): spins, one or two z1 images, two operand
// buffers, chain scalars, biases, [W1 itself when w1l], [the Philox hand-over area]
// split_operands: the 3 x bf16 split sampler's operand buffers ([8 k-steps][3 terms][64][4] dwords each)
inline size_t plan_sweep_lds_bytes(int N, int Hp, int n_hidden, bool w1l, bool rbm, int uh_floats = 0, bool split_operands = false) {
  const int Nst = (N + 3) & ~3, NT = Hp / 16, ZS = Hp + 4;
  const size_t xb = split_operands ? (size_t)8 * 3 * 256 : (size_t)NT * 256;
  return sizeof(float) * ((size_t)16 * Nst + (size_t)(w1l ? 1 : 2) * 16 * ZS + (size_t)2 * xb + 16 +
                          16 + 7 * 16 + Hp + (size_t)n_hidden * Hp + (rbm ? (w1l ? 0 : Nst) + 16 : 0) +
                          (w1l ? (size_t)N * (Hp + 4) : 0) + uh_floats);
}

// LDS the sampler needs at least (W1 streamed from L2); vmc_create rejects shapes beyond 160 KiB
inline size_t plan_sweep_lds_required(int N, int Hp, int n_hidden, bool rbm) {
  return plan_sweep_lds_bytes(N, Hp, n_hidden, false, rbm);
}

// which instantiation of k_sweep16<NT, NW, ...> a launch takes and with how much LDS
struct SweepPlan {
  int ok;          // 0: the shape does not fit (launch refused)
  int w1l;         // W1 held in LDS
  int fast;        // 0: general variant; 2 / 4: prefetched-Philox variant with UPRE = 2 / 4 draws per lane
  int uh_lds;      // the dedicated hand-over area is part of the LDS
  size_t lds;
};
inline SweepPlan plan_sweep(int N, int NT, int NW, int n_hidden, bool rbm, bool no_w1l, bool plain, bool tuned) {
  SweepPlan p;
  memset(&p, 0, sizeof(p));
  const int Hp = NT * 16;
  const size_t lds_full = plan_sweep_lds_bytes(N, Hp, n_hidden, true, rbm);
  // (more than 256 units: W1 alone would need > 160 KiB; those variants are not instantiated)
  const bool w1l = NT <= 16 && lds_full <= PLAN_LDS_PER_CU && !no_w1l;
  size_t lds = w1l ? lds_full : plan_sweep_lds_bytes(N, Hp, n_hidden, false, rbm);
  if (lds > PLAN_LDS_PER_CU) return p;
  const int nblk = (N + 3) / 4;
  const bool fast2 = nblk <= 32 && plain, fast4 = nblk <= 64 && plain;
  // five-slot hand-over area of the UPRE = 4 variant at 256 units (sweep16_body: UH_IN_X is false)
  if (NT == 16 && NW == 8 && !w1l && !fast2 && fast4) {
    const size_t with_uh = plan_sweep_lds_bytes(N, Hp, n_hidden, false, rbm, 4 * 5 * 256);
    if (with_uh <= PLAN_LDS_PER_CU) { lds = with_uh; p.uh_lds = 1; }
  }
  p.ok = 1; p.w1l = w1l ? 1 : 0; p.lds = lds;
  p.fast = !tuned ? 0 : (fast2 ? 2 : (fast4 ? 4 : 0));
  return p;
}

// k_sweep8 (sweep8.hip): eight chains per workgroup on the 4x4x1 MFMA shape, bit-identical chains to k_sweep16.
// fully_connected + relu, 128 or 256 padded units (one wave per 32 units), at least one H x H layer, one Philox site
// block per lane of a chain's group (8 * Hp / 32 lanes): n_sites <= Hp.  LDS: spins [8][Nst], two operand buffers
// [8][Hp + 16], three [8] int arrays, w_out, the biases of the H x H layers, [W1 when it fits].
struct Sweep8Plan {
  int ok;
  int w1l;
  size_t lds;
};
inline size_t plan_sweep8_lds_bytes(int N, int Hp, int n_hidden, bool w1l) {
  const size_t Nst = (size_t)((N + 3) & ~3);
  return sizeof(float) * (8 * Nst + 2 * 8 * ((size_t)Hp + 16) + 24 + (size_t)Hp + (size_t)n_hidden * Hp +
                          (w1l ? (size_t)N * ((size_t)Hp + 4) : 0));
}
inline Sweep8Plan plan_sweep8(int N, int Hp, int n_hidden, bool no_w1l) {
  Sweep8Plan p;
  memset(&p, 0, sizeof(p));
  if ((Hp != 128 && Hp != 256) || n_hidden < 1 || N < 2 || N > Hp || N > 256) return p;
  const size_t full = plan_sweep8_lds_bytes(N, Hp, n_hidden, true);
  const bool w1l = !no_w1l && full <= PLAN_LDS_PER_CU;
  const size_t lds = w1l ? full : plan_sweep8_lds_bytes(N, Hp, n_hidden, false);
  if (lds > PLAN_LDS_PER_CU) return p;
  p.ok = 1; p.w1l = w1l ? 1 : 0; p.lds = lds;
  return p;
}
// Chains per sampler workgroup: 8 when the shape has a k_sweep8 and sixteen-chain tiles would leave at least half of
// the CUs without one (graph_builders.py:57-88: the batch is the only parallel axis); forced: CGS_VMC_SWEEP_TILE
// (0 = this rule, 8, 16; 8 is honoured where a k_sweep8 exists).
inline int plan_sweep_tile(long long B, int num_cus, bool sweep8_ok, int forced) {
  if (!sweep8_ok || forced == 16) return 16;
  if (forced == 8) return 8;
  return (B + 15) / 16 <= num_cus / 2 ? 8 : 16;
}

// 384 or 512 padded units and at least one H x H layer; LDS of k_tail_lds: two operand buffers
// [2 halves][NT][64][4], partial dots, row meta, biases, w_out
inline size_t plan_tail_lds_bytes(int Hp, int n_hidden) {
  const size_t nt = (size_t)Hp / 16;
  return sizeof(float) * (2 * (2 * nt * 256) + 4 * 32 + 96 + (size_t)n_hidden * nt * 16 + nt * 16);
}
inline bool plan_tail_lds_supported(int Hp, int n_hidden) {
  return (Hp == 384 || Hp == 512) && n_hidden >= 1 && plan_tail_lds_bytes(Hp, n_hidden) <= PLAN_LDS_PER_CU;
}

// ------------------------------------------------------------------------------- vmc_create
struct DescPlan {
  int rbm, conv, resnet, one_d;
  int conv_general;          // conv beyond the fused kernels' limits (or forced): conv_general.hip
  int wide, wide_fast;       // > 256 units; of those, the fused 384 / 512-unit kernels
  int Hp;                    // padded units of the dense kernels (conv: 64, unused)
  int n_hh;                  // H x H layers
  long long P;               // parameters
  ParamLayout lay;
  ConvGeom cg;
};

// Everything vmc_create decides before it touches the device.  Returns a vmc_status; `msg` receives the
// reason.  wide_fast_allowed = false is CGS_VMC_WIDE_FAST=0.
// conv_general_pref (CGS_VMC_CONV_GENERAL): 1 the general convolution path for every shape, 0 where the fused kernels refuse
// the shape or the patch kernels beat them (plan_cgen_patch_routes), -1 only where the fused kernels refuse it
inline int plan_desc(const vmc_desc* d, bool wide_fast_allowed, DescPlan* out, char* msg, size_t msg_len,
                     int conv_general_pref = 0) {
  memset(out, 0, sizeof(*out));
#define PLAN_FAIL(code, text) do { snprintf(msg, msg_len, "%s", text); return code; } while (0)
  if (d->ansatz < VMC_ANSATZ_FULLY_CONNECTED || d->ansatz > VMC_ANSATZ_RES_NET_1D)
    PLAN_FAIL(VMC_ERR_UNSUPPORTED, "only the fully_connected, rbm, conv_1d/2d and res_net_1d/2d ansatz types have HIP kernels");
  const bool rbm = d->ansatz == VMC_ANSATZ_RBM;
  const bool conv = d->ansatz >= VMC_ANSATZ_CONV_2D;
  const bool resnet = d->ansatz == VMC_ANSATZ_RES_NET_2D || d->ansatz == VMC_ANSATZ_RES_NET_1D;
  const bool one_d = d->ansatz == VMC_ANSATZ_CONV_1D || d->ansatz == VMC_ANSATZ_RES_NET_1D;
  out->rbm = rbm; out->conv = conv; out->resnet = resnet; out->one_d = one_d;
  if (d->n_sites < 2 || d->batch_size < 1 || d->num_layers < ((rbm || resnet) ? 0 : 1) || d->layer_size < 1)
    PLAN_FAIL(VMC_ERR_INVALID, "n_sites >= 2, batch_size, layer_size >= 1, num_layers >= 1 (rbm, res_net_2d: >= 0) required");
  ConvGeom& cg = out->cg;
  if (conv) {
    // Conv2DNetwork reshapes its input to [-1, size_x, size_y, 1] (wavefunctions.py:596-597);
    // Conv1DNetwork expands [B, N] to [B, N, 1] (wavefunctions.py:511): an N x 1 lattice here
    const int sx = one_d ? d->n_sites : d->size_x, sy = one_d ? 1 : d->size_y;
    if (sx < 1 || sy < 1 || (long long)sx * sy != d->n_sites)
      PLAN_FAIL(VMC_ERR_INVALID, "size_x * size_y must equal num_sites");
    // Beyond the limits of the fused kernels (feature maps in LDS: kernel_size <= 9 -- one instantiation per size --,
    // num_conv_filters <= 64 -- four channel blocks of 16 --, a sample's two maps within 160 KiB) the general path
    // of conv_general.hip serves: feature maps in HBM, a convolution = im2col (explicit, or in the A operand's address)
    // + one GEMM (round 5: forward, local energies, sampler, the gradient accumulators of both optimizers and
    // stochastic reconfiguration through the one-call solves)
    bool general = conv_general_pref > 0;
    if (d->kernel_size < 1 || d->kernel_size > CONV_GENERAL_MAX_K)
      PLAN_FAIL(VMC_ERR_UNSUPPORTED, "kernel_size 1..31 supported by the convolution kernels");
    if (d->layer_size > CONV_GENERAL_MAX_F)
      PLAN_FAIL(VMC_ERR_UNSUPPORTED, "num_conv_filters > 1024 not supported by the convolution kernels");
    if (d->kernel_size > CONV_MAX_K || d->layer_size > CONV_FP * CONV_MAX_NCB) general = true;
    if (sx < d->kernel_size / 2 || (!one_d && sy < d->kernel_size / 2) || sx > 1023 || sy > 1023)
      PLAN_FAIL(VMC_ERR_UNSUPPORTED, "lattice sides must be in [kernel_size / 2, 1023]");
    if ((long long)d->num_layers > CONV_MAX_LAYERS)
      PLAN_FAIL(VMC_ERR_UNSUPPORTED, "too many convolutions");
    cg.K = d->kernel_size; cg.D1 = sx; cg.D2 = sy; cg.N = d->n_sites; cg.F = d->layer_size;
    cg.n_conv = resnet ? 1 + 2 * d->num_layers : d->num_layers;
    cg.resnet = resnet ? 1 : 0; cg.hact = d->nonlinearity;
    cg.GS = (4 * cg.N + 63) / 64 * 64;
    cg.NCB = (cg.F + CONV_FP - 1) / CONV_FP;
    cg.CS = 4 * cg.NCB * cg.GS;
    if (one_d) {   // layers.py:66-72: k/2 in front, k - 1 - k/2 behind (odd k: (k-1)/2 both)
      cg.KW = 1; cg.lo = cg.K / 2; cg.hi = cg.K - 1 - cg.lo; cg.lo2 = cg.hi2 = 0;
    } else {       // layers.py:132-141: (k-1)/2 in front, k/2 behind, both axes
      cg.KW = cg.K; cg.lo = cg.lo2 = (cg.K - 1) / 2; cg.hi = cg.hi2 = cg.K / 2;
    }
    if (cg.n_conv > CONV_MAX_LAYERS) PLAN_FAIL(VMC_ERR_UNSUPPORTED, "too many convolutions");
    if ((long long)d->batch_size * cg.CS >= (1LL << 31)) general = true;   // 32-bit tape offsets of the fused kernels
    if (plan_conv_rows_lds(cg, 1) > PLAN_LDS_PER_CU) general = true;       // a sample's maps beyond 160 KiB of LDS
    if (!general && conv_general_pref == 0 && plan_cgen_patch_routes(cg, d->batch_size)) general = true;   // a lattice much wider than the network's reach
    if (general && (long long)cg.N * plan_cgen_lda(cg) >= (1LL << 28))
      PLAN_FAIL(VMC_ERR_UNSUPPORTED, "lattice x kernel x filters too large for the general convolution path (one sample's im2col rows beyond 1 GiB)");
    out->conv_general = general ? 1 : 0;
  }
  if (d->nonlinearity < 0 || d->nonlinearity > 6 || d->output_activation < 0 || d->output_activation > 6)
    PLAN_FAIL(VMC_ERR_INVALID, "unknown activation id (layers.NONLINEARITIES has 7 entries)");
  if (rbm && d->output_activation != VMC_ACT_EXP)
    PLAN_FAIL(VMC_ERR_INVALID, "the rbm ansatz has no output_activation: it is always exp (wavefunctions.py:419-420)");
  const bool wide = !conv && d->layer_size > 256;
  const int n_hh = conv ? 0 : (rbm ? d->num_layers : d->num_layers - 1);
  // 257 .. 512 units run the fused kernels (every activation); beyond that -- or with the fused path
  // switched off or out of LDS -- the general path (every activation: its back-propagation reads f' off
  // the activation, or off the stored f'(z) for the cosine)
  bool wide_fast = false;
  if (wide && d->layer_size <= 512) {
    const int hp = (d->layer_size + 127) / 128 * 128;     // 8 waves x whole 16-unit tiles
    wide_fast = wide_fast_allowed && (n_hh == 0 || plan_tail_lds_supported(hp, n_hh)) &&
                plan_sweep_lds_required(d->n_sites, hp, n_hh, rbm) <= PLAN_LDS_PER_CU;
  }
  if (wide && d->layer_size > 4096) PLAN_FAIL(VMC_ERR_UNSUPPORTED, "fc_layer_size > 4096 is not supported");
  if (!conv && !wide) {  // the sampler keeps 16 chains' spins, z1 and operands in LDS (160 KiB per CU)
    const int hp = (d->layer_size + 63) / 64 * 64;
    const size_t need = plan_sweep_lds_required(d->n_sites, hp, n_hh, rbm);
    if (need > PLAN_LDS_PER_CU) {
      snprintf(msg, msg_len, "num_sites = %d with %d hidden units needs %zu bytes of LDS for the sampler's "
               "chain state (limit 163840)", d->n_sites, d->layer_size, need);
      return VMC_ERR_UNSUPPORTED;
    }
  }
#undef PLAN_FAIL
  out->wide = wide; out->wide_fast = wide_fast; out->n_hh = n_hh;
  out->Hp = conv ? 64 : (wide_fast ? (d->layer_size + 127) / 128 * 128 : (d->layer_size + 63) / 64 * 64);
  if (conv) {
    out->P = plan_num_params_conv(cg.n_conv, cg.F, (long long)cg.K * cg.KW);
    out->lay = plan_layout(false, d->n_sites, d->layer_size, 1);     // minimal dense-side shapes (unused)
    out->n_hh = 0;
  } else {
    out->P = plan_num_params_dense(d->ansatz, d->n_sites, d->layer_size, d->num_layers);
    out->lay = plan_layout(rbm, d->n_sites, d->layer_size, d->num_layers);
  }
  if (msg_len) msg[0] = 0;
  return VMC_OK;
}

// ------------------------------------------------------------------------------- weight-gradient GEMM
#define WG_TM 64           // output tile of the batched weight-gradient kernel (k_wgrad, grad.hip)
#define WG_TN 64
#define WG_TK 32
#define WG_MAX_SPLIT 32    // workspace bound

// One problem of the batched weight-gradient launch: C[m_rows (+ ones row)][n_cols] over K samples
struct WgradShape { int m_rows, n_cols; };

// tiles of one problem (the N = 1 output / onsite problem is a one-column tile row like any other: its
// MFMAs multiply mostly zeros, but it needs no code of its own -- a VALU column sum over all samples in a
// workgroup of its own was latency bound at 190 us)
PLAN_HD inline int plan_wgrad_tiles(int m_rows, int n_cols) {
  return ((m_rows + WG_TM - 1) / WG_TM) * ((n_cols + WG_TN - 1) / WG_TN);
}

// K slices: as many as fill the CUs once (tiles x slices + the other blocks of the launch <= num_cus), each a
// whole number of WG_TK steps, at least 64 samples per slice.  forced > 0: the measurement knob
// CGS_VMC_WGRAD_SLICES
inline int plan_wgrad_slices(long long total_tiles, long long K, int num_cus, int other_blocks = 0, int forced = 0) {
  if (total_tiles <= 0) return 1;
  long long s = (num_cus - other_blocks) / total_tiles;
  if (forced > 0) s = forced;
  const long long by_k = (K + 63) / 64;
  if (s > by_k) s = by_k;
  if (s > WG_MAX_SPLIT) s = WG_MAX_SPLIT;
  if (s < 1) s = 1;
  // whole k-steps per slice; then no more slices than have samples (no slice is ever empty)
  long long kc = (K + s - 1) / s;
  kc = (kc + WG_TK - 1) / WG_TK * WG_TK;
  s = (K + kc - 1) / kc;
  return s < 1 ? 1 : (int)s;
}

// samples per slice, rounded up to whole k-steps
PLAN_HD inline int plan_wgrad_kchunk(int K, int slices) {
  int kc = (K + slices - 1) / slices;
  return (kc + WG_TK - 1) / WG_TK * WG_TK;
}

// XCD-aware block order: workgroups go to the 8 XCDs round robin by linear block id and every XCD has
// its own L2.  The W = tiles x slices work items, slice-major (all tiles of a k-slice share that slice's
// operand rows), are cut into 8 contiguous ranges of G = ceil(W / 8), one per XCD: every XCD gets the
// same number of workgroups (slices per XCD would leave XCDs idle whenever slices % 8 != 0: 5 slices
// ran on 5 of 8 XCDs) and mostly one slice's rows in its L2.  Block b -> item (b % 8) G + b / 8.
struct WgradBlock { int slice, tile; };
PLAN_HD inline WgradBlock plan_wgrad_block(int block, int tiles, int slices) {
  WgradBlock b;
  const int W = tiles * slices, G = (W + 7) / 8;
  const int j = block >> 3, w = (block & 7) * G + j;
  if (j >= G || w >= W) { b.slice = -1; b.tile = 0; return b; }
  b.slice = w / tiles;
  b.tile = w % tiles;
  return b;
}
inline int plan_wgrad_grid(int tiles, int slices) { return tiles <= 0 ? 0 : 8 * ((tiles * slices + 7) / 8); }

// MFMA tiles of the whole dense gradient launch: the output (FC) / onsite (RBM) problem
// [k_in = H or N][1], n_hh hidden problems [H][H], the first layer [N][H]
// out_in_tiles: the N = 1 layer (w_out, b_out / w_on, b_on) is one of the tile problems; false: its sums come
// from the back-propagation kernel's per-workgroup partials and are folded by plan_wgrad_fold_blocks(H)
// workgroups of the same launch (fully_connected on the fused kernels)
inline int plan_wgrad_total_tiles(int N, int H, int n_hh, bool rbm, bool out_in_tiles = true) {
  return (out_in_tiles ? plan_wgrad_tiles(rbm ? N : H, 1) : 0) + n_hh * plan_wgrad_tiles(H, H) + plan_wgrad_tiles(N, H);
}
#define WG_FOLD_OUT 128    // outputs (two sums x (H weights + bias)) per fold workgroup, four row groups each
inline int plan_wgrad_fold_blocks(int H) { return (2 * (H + 1) + WG_FOLD_OUT - 1) / WG_FOLD_OUT; }

// floats of the partial-tile workspace: [tiles][slices][2 x WG_TM x WG_TN + 2 x WG_TN]
inline long long plan_wgrad_ws_floats(long long tiles, int slices) {
  return tiles * slices * 2 * ((long long)WG_TM * WG_TN + WG_TN);
}

// ------------------------------------------------------------------------------- stochastic reconfiguration
#define RD_MAXB 16         // problems per row-dot launch
#define RD_TM 128          // samples per row-dot tile

// dispatch order of the row-dot problems of one launch: largest K first (list scheduling: the light
// tiles fill the end), stable; first_tile[j] = first tile of the j-th dispatched problem
inline int plan_rowdot_schedule(const int* K, const int* M, int n, int* order, int* first_tile) {
  for (int j = 0; j < n; ++j) order[j] = j;
  for (int a = 1; a < n; ++a)
    for (int b = a; b > 0 && K[order[b]] > K[order[b - 1]]; --b) {
      const int tmp = order[b]; order[b] = order[b - 1]; order[b - 1] = tmp;
    }
  int tiles = 0;
  for (int j = 0; j < n; ++j) {
    first_tile[j] = tiles;
    tiles += (M[order[j]] + RD_TM - 1) / RD_TM;
  }
  first_tile[n] = tiles;
  return tiles;
}

inline int plan_sr_wsum_slices(int R, int num_cus) {
  int s = (R + 63) / 64;              // at least 64 samples per slice
  if (s > num_cus) s = num_cus;
  return s < 1 ? 1 : s;
}
