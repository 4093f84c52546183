// Convolutional ansatz types (Conv2DNetwork, wavefunctions.py:531-615; ResNet2D, 710-809; on
// layers.Conv2dPeriodic / ResBlock2d, layers.py:89-229) for gfx950: shared descriptors and
// launcher declarations.  Kernels: conv.hip.  DESIGN.md 4 "Convolutional ansatz".
#pragma once
#include "common.hpp"

// (ConvGeom, CONV_FP / CONV_MAX_NCB / CONV_MAX_LAYERS and every LDS / grid formula: plan.hpp)

// One packed parameter set (k_conv_pack); co / ci = output / input channel block of 16:
//   w0   [NCB co][Q0][64]   first convolution (1 input channel), taps are the k index:
//                           lane (m, g) of k-step q = W0[tap 4q+g][0][cout 16 co + m]
//   wf   [n_conv-1][NCB co][NCB ci][K*KW][64][4] forward fragments of convolutions 1..: lane (m, g),
//                           element e = W[tap][cin 16 ci + 4g+e][cout 16 co + m]
//                           (A operand of v_mfma_f32_16x16x4_f32)
//   wb   same shape, the transposed convolution: block (co, ci) of the image produces channel block
//                           co of d/d(input) from block ci of d/d(output): lane (m, g), e =
//                           W[K*K-1-tap][cin 16 co + m][cout 16 ci + 4g+e]
//   bias [n_conv][16 NCB]
struct ConvParams {
  const float* w0;
  const float* wf;
  const float* wb;
  const float* bias;
};

struct ConvRowsArgs {
  ConvGeom g;
  ConvParams p;
  const float* configs;     // [n_base][N] +-1
  const int2* rowinfo;      // [n_rows] {chain, +-(bond+1) or 0}
  const int2* bonds;        // [n_bonds]
  const float* half_jx;     // [n_bonds]
  const float* logit_base;  // [n_base] (ratio mode)
  const int* n_rows_dev;    // device row count or nullptr
  int n_rows;               // host row count / upper bound
  int ratio;                // 1: out = half_jx[bond] * psi'/psi, 0: out = logit
  int oact;
  int G;                    // samples per pass (LDS resident)
  float* out;               // [n_rows]
  float* tape;              // [n_conv-1][n_rows][CS] inputs of convolutions 1.. (gradient path; the cosine
                            // tapes the PRE-activation, its derivative needs z) or nullptr
  long long tape_stride;    // floats between the tapes of consecutive convolutions
};

struct ConvSweepArgs {
  ConvGeom g;
  ConvParams p;
  const float* configs_in;  // [B][N]
  const float* logit_in;    // [B] (read when cache_in_valid)
  float* configs;           // [B][N] out
  float* logit;             // [B] out
  unsigned long long* accepted;
  const int* inj_up; const int* inj_dn; const float* inj_u;
  unsigned char* acc_mask;
  int* dbg_up; int* dbg_dn; float* dbg_u;
  int oact;
  int cache_in_valid;
  int B, G;
  int chain_offset;
  uint32_t seed_lo, seed_hi;
  unsigned long long step0;
  long long n_steps;
};

struct ConvBackArgs {
  ConvGeom g;
  ConvParams p;
  const float* tape;        // [n_conv-1][B][CS]
  long long tape_stride;
  const float* oscale;      // [B] (1/psi) d psi / d logit (1 for the exp output)
  float* delta;             // [n_conv][B][CS]  d logit / d (output of convolution l)
  long long delta_stride;
  int B, G;
};

struct ConvDwArgs {
  ConvGeom g;
  const float* configs;     // [B][N]
  const float* tape;  long long tape_stride;
  const float* delta; long long delta_stride;
  const float* w;           // [B] weights of the second (scaled) sum
  int B;
  int n_slices;             // sample slices = gridDim.x
  float* ws;                // [n_slices][n_conv][2][(K*K*16 NCB + 1) * 16 NCB] partial sums
  float* g1; float* g2;     // accumulators (theta layout), += on reduce; g1 == nullptr: weighted sum only
  int band_rows;            // lattice rows staged at a time (set by launch_conv_dw: all of them when LDS allows)
};

// stochastic reconfiguration (extension): t[row] = O_row . p over the stored samples
struct ConvSrRowdotArgs {
  ConvGeom g;
  ConvParams p;             // the CG direction p packed like a parameter set (launch_conv_pack)
  const float* configs;     // [n_rows][N] stored chains
  const float* tape;  long long tape_stride;     // [n_conv-1][R][CS]
  const float* delta; long long delta_stride;    // [n_conv][R][CS]
  float* t;                 // [n_rows]
  int n_rows, G;
};

size_t conv_rows_lds(const ConvGeom& g, int G);
size_t conv_lds_cap(const ConvGeom& g);   // LDS budget of one workgroup (two per CU when a sample fits)
int conv_waves(const ConvGeom& g);        // waves per workgroup of the conv kernels (4; two channel blocks: 8)
int conv_pick_group(const ConvGeom& g, int waves);
int conv_pick_sweep_group(const ConvGeom& g, long long B, int num_cus, int waves);   // chains per sampler workgroup
long long conv_num_params(int n_conv, int F, int taps);
hipError_t launch_conv_pack(hipStream_t s, const float* theta, const ConvGeom& g, float* w0,
                            float* wf, float* wb, float* bias);
hipError_t launch_conv_rows(hipStream_t s, const ConvRowsArgs& a, int num_cus);
hipError_t launch_conv_sweep(hipStream_t s, const ConvSweepArgs& a);
hipError_t launch_conv_back(hipStream_t s, const ConvBackArgs& a, int num_cus);
hipError_t launch_conv_dw(hipStream_t s, const ConvDwArgs& a);
hipError_t launch_conv_sr_rowdot(hipStream_t s, const ConvSrRowdotArgs& a, int num_cus);

// ---- general path (conv_general.hip): feature maps in HBM [row][site][Fp], a convolution = im2col + one GEMM
#define CGEN_PRE_SELU 7                                            // pre_act ids: -1 none, 0 .. 6 = VMC_ACT_*, 7 selu
inline int cgen_fp(const ConvGeom& g) { return (g.F + 3) & ~3; }
inline int cgen_kdim(const ConvGeom& g, int l) { return g.K * g.KW * (l == 0 ? 1 : g.F); }
inline long long cgen_off_w(const ConvGeom& g, int l) {            // theta: w[k][kw][Cin][F] then b[F], creation order
  long long o = 0;
  for (int i = 0; i < l; ++i) o += (long long)cgen_kdim(g, i) * g.F + g.F;
  return o;
}
inline long long cgen_off_b(const ConvGeom& g, int l) { return cgen_off_w(g, l) + (long long)cgen_kdim(g, l) * g.F; }
struct CgenIm2colArgs {
  ConvGeom g;
  int layer;                 // 0: gather from the spins (with the exchanged pair negated); > 0: from a feature map
  const float* src;          // layer 0: configs [n_base][N]; else the input map [rows][N][Fp]
  int Fp;
  int pre_act;               // applied to the gathered values (layer > 0)
  int inverse;               // layer > 0: gather of the TRANSPOSED convolution (the position whose tap t reads the site)
  const int2* rowinfo;       // layer 0: [..] {chain, +-(bond+1) or 0}, or nullptr: row r is chain row0 + r
  long long row0;            // first row of this block in rowinfo / iup / idn
  const int2* bonds;
  const int* iup; const int* idn;   // layer 0: the proposed exchange of row r (or nullptr)
  int rows;                  // rows of this block
  int lda;                   // floats per row of A
  float* A;                  // [rows * N][lda]
};
hipError_t launch_cgen_im2col(hipStream_t s, const CgenIm2colArgs& a);
// One convolution of the general path at <= 16 filters WITHOUT an im2col matrix (conv_band.hip): bands of lattice rows
// staged through LDS with their periodic halo, 16 output channels x 16 positions per MFMA tile, weights in registers.
struct CgenBandArgs {
  ConvGeom g;
  int layer;                 // 0: the spins (exchanged pair negated); > 0: a feature map
  int Fp;
  const float* w;            // the convolution's weights as they lie in theta: [K][KW][Cin][F]
  const float* bias;         // [F]
  const float* in;           // layer > 0: [rows][N][Fp]
  float* out;                // [rows][N][Fp]
  int rows;
  int pre_act;               // applied to the staged input (layer > 0; -1: none)
  int epilogue, act;         // GemmArgs' ids: 1 f(v + bias), 4 v + bias, 8 C + v + bias, 11 selu(v + bias)
  const float* configs; const int2* rowinfo; long long row0; const int2* bonds; const int* iup; const int* idn;   // layer 0 (as CgenIm2colArgs)
  int band_rows;             // set by the launcher
};
bool cgen_band_ok(const ConvGeom& g);
int cgen_band_rows(const ConvGeom& g);
hipError_t launch_cgen_band(hipStream_t s, const CgenBandArgs& a, int num_cus);
// the FIRST convolution (one input channel) of any filter count without an im2col matrix (k_cgen_first_direct)
bool cgen_first_direct_ok(const ConvGeom& g, int epilogue);
hipError_t launch_cgen_first_direct(hipStream_t s, const CgenBandArgs& a, int num_cus);
hipError_t launch_cgen_rowsum(hipStream_t s, const float* fm, int rows, int N, int F, int Fp, double* out);
// The patch sampler (conv_patch.hip): n_steps exchange steps of every chain in ONE launch, one workgroup per chain; per
// step the two boxes of every convolution that the exchanged pair reaches are recomputed from the chain's stored maps
struct CgenPatchArgs {
  ConvGeom g;
  int Fp;
  const float* theta;        // the parameters as they lie: per convolution w[K][KW][Cin][F], b[F]
  float* maps;               // [n_conv][B][N][Fp]: the maps of the chains as they stand (cgen_post says what a map holds)
  long long map_stride;      // floats between two convolutions' maps (B N Fp)
  int post;                  // 1: maps hold activations (the epilogue applies them); 0: pre-activations (applied on the gather)
  int act, oact;
  float* configs; float* logit; const int* iup; const int* idn; const float* u;   // the chains; the first step's proposals
  unsigned long long* accepted;
  int B;
  uint32_t seed_lo, seed_hi; int chain_offset; unsigned long long step0; long long n_steps;
  unsigned long long* prof;  // diagnostic (CGS_VMC_CONV_PATCH_PROF=1): six phase clocks of chain 0, or nullptr
  // the local energies' form (launch_cgen_patch_rows): rows (chain, +-(bond + 1) or 0) of a row list instead of steps
  const int2* rowinfo; const int2* bonds; long long row0; long long n_rows; double* out_sum;
};
bool cgen_patch_ok(const ConvGeom& g, long long B);
hipError_t launch_cgen_patch_sweep(hipStream_t s, const CgenPatchArgs& a);
hipError_t launch_cgen_patch_rows(hipStream_t s, const CgenPatchArgs& a, int num_cus);
// map sum + candidate logit + Metropolis test / commit + the next step's proposal, one workgroup per chain (k_cgen_step_tail)
hipError_t launch_cgen_step_tail(hipStream_t s, const float* fm, int N, int F, int Fp, float* configs, float* logit, int B,
                                 int oact, int* iup, int* idn, float* u, unsigned long long* accepted, uint32_t seed_lo,
                                 uint32_t seed_hi, int chain_offset, unsigned long long next_step, bool do_propose);
hipError_t launch_cgen_accept(hipStream_t s, float* configs, float* logit, const float* lnew, const int* iup,
                              const int* idn, const float* u, int B, int N, int oact, unsigned long long* accepted,
                              unsigned char* acc_mask);
// gradient path: d logit / d (last map), d (.) f'(z), per-position weights, transposed weight image [T F][F] of a layer >= 1
hipError_t launch_cgen_fill(hipStream_t s, float* gm, const float* oscale, long long row0, int rows, int N, int F, int Fp);
hipError_t launch_cgen_dact(hipStream_t s, const float* d, const float* z, int pre, bool from_act, long long n, int F, int Fp,
                            float* out);
hipError_t launch_cgen_wpos(hipStream_t s, const float* w, long long row0, int rows, int N, float* wpos);
hipError_t launch_cgen_pack_t(hipStream_t s, const float* w, int T, int F, float* wt);
// SR (single rank): t[r] (+)= < y[r] , g[r] > in double; t = (float) td; the mean of t and sum (t - mean); weights t - mean
hipError_t launch_cgen_pairdot(hipStream_t s, const float* y, const float* gm, int rows, int N, int F, int Fp, double* t, bool first);
hipError_t launch_cgen_tstore(hipStream_t s, const double* td, int rows, float* t);
hipError_t launch_cgen_tmean(hipStream_t s, const float* t, int n, float* centre, float* usum);
hipError_t launch_cgen_tcentre_global(hipStream_t s, const float* t, int n, const float* count, float* centre, float* usum);
hipError_t launch_cgen_wpos_centred(hipStream_t s, const float* t, const float* centre, long long row0, int rows, int N, float* wpos);
inline long long cgen_off_wt(const ConvGeom& g, int l) { return (long long)(l - 1) * g.K * g.KW * g.F * g.F; }   // l >= 1
