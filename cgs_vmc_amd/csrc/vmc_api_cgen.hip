// The general convolution path (conv_general.hip / conv_band.hip kernels): forward, gradient and SR machinery of a ctx
// whose shape the fused convolution kernels refuse (vmc_debug_kernel_path == 6).  Split out of vmc_api.hip in round 6.
#include "vmc_ctx.hpp"

using namespace vmcapi;

namespace vmcapi {

#define CGEN_SPLITK 32     // K slices of the weight-gradient products of the general convolution path

// One convolution of the general path over `rows` row configurations: im2col gather of its input into cg_A, then the
// product with the parameter slice; residual: dst += (ResBlock2d's `v + h`).
// What a stored map of the general path holds: the ACTIVATION of a convolution's output (conv_plain: f(z_l) behind every
// convolution but the last; residual blocks: selu(u) behind a block's first convolution, the linear h elsewhere) -- so
// that the next convolution can gather it as it stands -- unless the hidden activation is the cosine, whose derivative
// needs the pre-activation: then the map holds z_l and f is applied on the gather (as in the first form of this path).
static bool cgen_post(const vmc_ctx* c) { return c->cg.resnet || c->cg.hact != VMC_ACT_COS_; }
static int cgen_in_pre(const vmc_ctx* c, int l) {       // activation applied while convolution l's input is gathered
  if (l == 0 || c->cg.resnet || cgen_post(c)) return -1;
  return c->cg.hact;
}
static bool cgen_implicit_on() {      // read per call: explicit against implicit gathers in one process (tests)
  const char* e = getenv("CGS_VMC_CONV_GENERAL_IMPLICIT");
  return !(e && atoi(e) == 0);
}
// The im2col matrix [cg_rows * N][plan_cgen_lda] (up to 768 MB) exists from the first launch that writes one: a ctx whose
// convolutions all take the band kernel or the implicit gather never allocates it for the forward and the sampler.
static int cgen_need_A(vmc_ctx* c) {
  if (c->cg_A) return VMC_OK;
  HIPCHK(c, dalloc(&c->cg_A, c->cg_rows * c->cg.N * plan_cgen_lda(c->cg)));
  return VMC_OK;
}

// CGS_VMC_CONV_BAND=0: the im2col + GEMM form for every filter count (read per call: A/B tests in one process)
static bool cgen_band_on() { const char* e = getenv("CGS_VMC_CONV_BAND"); return !(e && atoi(e) == 0); }

// the stream of the forward launches: `stream`, or the running sampler group's (run_sweep_cgen)
static hipStream_t cg_s(const vmc_ctx* c) { return c->cg_stream_cur ? c->cg_stream_cur : c->stream; }

static int cgen_conv(vmc_ctx* c, const ParamSet& p, const float* configs, const int2* rowinfo, const int* iup,
                     const int* idn, int l, int rows, const float* in, float* dst, long long row0) {
  const ConvGeom& g = c->cg;
  const int Fp = cgen_fp(g), lda = plan_cgen_lda(g);
  GemmArgs m; memset(&m, 0, sizeof(m));
  m.B = p.theta + cgen_off_w(g, l); m.sbk = g.F; m.sbn = 1;
  m.M = rows * g.N; m.N = g.F; m.K = cgen_kdim(g, l); m.C = dst; m.ldc = Fp;
  m.bias = p.theta + cgen_off_b(g, l); m.splitk = 1;
  if (!g.resnet) { m.epilogue = (l + 1 < g.n_conv && cgen_post(c)) ? 1 : 4; m.act = g.hact; }
  else m.epilogue = l == 0 ? 4 : ((l & 1) ? 11 : 8);       // initial convolution; selu(first_conv(h)); h + second_conv(.)
  const int pre = cgen_in_pre(c, l);
  // up to 16 filters: the band kernel (conv_band.hip) -- no im2col matrix, no 64-column tile for 16 columns
  const bool first_direct = l == 0 && cgen_band_on() && cgen_first_direct_ok(g, m.epilogue);
  if (cgen_band_on() && (cgen_band_ok(g) || first_direct)) {
    CgenBandArgs b; memset(&b, 0, sizeof(b));
    b.g = g; b.layer = l; b.Fp = Fp; b.w = m.B; b.bias = m.bias; b.in = in; b.out = dst; b.rows = rows;
    b.pre_act = pre; b.epilogue = m.epilogue; b.act = m.act;
    if (l == 0) {
      b.configs = configs; b.rowinfo = rowinfo; b.row0 = row0; b.bonds = c->bonds ? c->bonds : c->bond_dummy;
      b.iup = iup; b.idn = idn;
    }
    if (cgen_band_ok(g)) HIPCHK(c, launch_cgen_band(cg_s(c), b, c->num_cus));
    else HIPCHK(c, launch_cgen_first_direct(cg_s(c), b, c->num_cus));     // more than 16 filters: the first convolution only
    return VMC_OK;
  }
  // the gather inside the product's A operand (k_gemm_ring<., true>): no im2col matrix for this convolution
  if (l > 0 && pre < 0 && cgen_implicit_on()) {
    m.A = in; m.conv_a = 1; m.ca_N = g.N; m.ca_D1 = g.D1; m.ca_D2 = g.D2; m.ca_KW = g.KW; m.ca_lo = g.lo; m.ca_lo2 = g.lo2;
    m.ca_F = g.F; m.ca_Fp = Fp;
    if (gemm_conv_a_ok(m)) { HIPCHK(c, launch_gemm(cg_s(c), m)); return VMC_OK; }
    m.conv_a = 0;
  }
  PROPAGATE(cgen_need_A(c));
  CgenIm2colArgs a;
  memset(&a, 0, sizeof(a));
  a.g = g; a.layer = l; a.Fp = Fp; a.pre_act = pre; a.rows = rows; a.lda = lda; a.A = c->cg_A + c->cg_map_row0 * g.N * lda;
  if (l == 0) {
    a.src = configs; a.rowinfo = rowinfo; a.row0 = row0; a.bonds = c->bonds ? c->bonds : c->bond_dummy;
    a.iup = iup; a.idn = idn;
  } else {
    a.src = in;
  }
  HIPCHK(c, launch_cgen_im2col(cg_s(c), a));
  m.A = c->cg_A + c->cg_map_row0 * g.N * lda; m.sam = lda; m.sak = 1;
  HIPCHK(c, launch_gemm(cg_s(c), m));
  return VMC_OK;
}

// an untaped forward of n_rows rows is ONE block (its last map is then still in place behind it: cgen_last_map)
bool cgen_single_block(const vmc_ctx* c, long long n_rows) {
  const long long blk_rows = (cgen_band_on() && cgen_band_ok(c->cg)) ? c->cg_rows_fwd : c->cg_rows;
  return n_rows <= blk_rows;
}
const float* cgen_last_map(const vmc_ctx* c) {
  return (c->cg.resnet ? c->cg_fm[0] : c->cg_fm[(c->cg.n_conv - 1) & 1]) + c->cg_map_row0 * c->cg.N * cgen_fp(c->cg);
}

// The patch kernels (conv_patch.hip) -- 0: not for this ctx / switched off; 1: where they pay; 2: wherever the shape allows.
// CGS_VMC_CONV_PATCH=0 | 2, read per call (tests compare the two forms in one process); a box has the BAND kernel's bits, so not
// with CGS_VMC_CONV_BAND=0.
int cgen_patch_mode(const vmc_ctx* c) {
  if (!c->conv_general) return 0;
  const char* pe = getenv("CGS_VMC_CONV_PATCH");
  const int mode = pe ? atoi(pe) : 1;
  if (mode == 0 || !cgen_band_on() || !cgen_patch_ok(c->cg, c->B)) return 0;
  return mode == 2 ? 2 : 1;
}
// the maps of every convolution of the ctx's B chains at parameter set `which`: one taped forward, blocks of the im2col-sized rows
int cgen_patch_maps(vmc_ctx* c, int which) {
  const ConvGeom& g = c->cg;
  const long long map_floats = (long long)c->B * g.N * cgen_fp(g);
  if (!c->cg_pmaps) HIPCHK(c, dalloc(&c->cg_pmaps, g.n_conv * map_floats));
  for (long long r0 = 0; r0 < c->B; r0 += c->cg_rows) {
    const long long rows = c->B - r0 < c->cg_rows ? c->B - r0 : c->cg_rows;
    PROPAGATE(cgen_forward(c, which, c->configs, nullptr, rows, nullptr, nullptr, false, nullptr,
                           c->cg_pmaps + r0 * g.N * cgen_fp(g), map_floats, r0));
  }
  return VMC_OK;
}
void cgen_patch_args(const vmc_ctx* c, int which, CgenPatchArgs* a) {
  const ConvGeom& g = c->cg;
  memset(a, 0, sizeof(*a));
  a->g = g; a->Fp = cgen_fp(g); a->theta = c->ps[which].theta; a->maps = c->cg_pmaps;
  a->map_stride = (long long)c->B * g.N * cgen_fp(g);
  a->post = (g.resnet || g.hact != VMC_ACT_COS_) ? 1 : 0; a->act = g.hact; a->oact = c->oact;      // cgen_post
  a->configs = c->configs; a->B = c->B;
}

// tape != nullptr (gradient path, n_rows <= cg_rows): the map of convolution l is kept at tape + l * tape_stride
// (cgen_post says what it holds; for the second convolution of a residual block the block's output h + v)
int cgen_forward(vmc_ctx* c, int which, const float* configs, const int2* rowinfo, long long n_rows,
                 const int* iup, const int* idn, bool ratio, float* out, float* tape, long long tape_stride,
                 long long first_row) {
  const ConvGeom& g = c->cg;
  const ParamSet& p = c->ps[which];
  const int Fp = cgen_fp(g);
  auto conv = [&](int l, int rows, const float* in, float* dst, long long row0) -> int {
    return cgen_conv(c, p, configs, rowinfo, iup, idn, l, rows, in, dst, row0);
  };
  const long long moff = c->cg_map_row0 * g.N * Fp;      // (a sampler group's slice of the maps; 0 elsewhere)
  auto map = [&](int l) { return tape ? tape + (long long)l * tape_stride : c->cg_fm[g.resnet ? (l & 1 ? 1 : 0) : (l & 1)] + moff; };
  if (tape && n_rows > c->cg_rows) return fail(c, VMC_ERR_STATE, "taped forward beyond one block");
  // The rows of a row list over the ctx's chains -- the local energies' connected configurations (operators.py:162-169:
  // the chain with one antiparallel bond exchanged) -- through the patch kernel: the chains' maps once (B forwards), then
  // per row the boxes around the bond's two sites and the last map's sum with them overlaid; the sums are k_cgen_rowsum's
  // bits, the rest of the block is the same (k_wide_out_part).
  if (!tape && out && !iup && rowinfo && rowinfo != c->rowinfo_id && configs == c->configs && n_rows > 0) {
    const int mode = cgen_patch_mode(c);
    if (mode == 2 || (mode == 1 && plan_cgen_patch_pays(g) && n_rows >= 4LL * c->B)) {
      PROPAGATE(cgen_patch_maps(c, which));
      CgenPatchArgs a;
      cgen_patch_args(c, which, &a);
      a.rowinfo = rowinfo; a.bonds = c->bonds ? c->bonds : c->bond_dummy; a.out_sum = c->cg_sum;
      const long long blk = c->cg_rows_fwd;
      for (long long blk0 = 0; blk0 < n_rows; blk0 += blk) {
        const long long rows = n_rows - blk0 < blk ? n_rows - blk0 : blk;
        a.row0 = first_row + blk0; a.n_rows = rows;
        HIPCHK(c, launch_cgen_patch_rows(cg_s(c), a, c->num_cus));
        const WideOnsite on{nullptr, nullptr, nullptr, nullptr, nullptr};
        HIPCHK(c, launch_wide_out_part(cg_s(c), c->cg_sum, 1, c->cg_zero, (int)rows, rowinfo, a.row0,
                                       c->half_jx, p.logit, c->oact, ratio, out, on));
      }
      return VMC_OK;
    }
  }
  // block size: the im2col-sized one, or -- untaped, every convolution on the band kernel -- the maps-sized one
  const long long blk_rows = (!tape && cgen_band_on() && cgen_band_ok(g)) ? c->cg_rows_fwd : c->cg_rows;
  for (long long blk0 = 0; blk0 < n_rows; blk0 += blk_rows) {
    const long long row0 = first_row + blk0;
    const int rows = (int)(n_rows - blk0 < blk_rows ? n_rows - blk0 : blk_rows);
    const float* last;
    PROPAGATE(conv(0, rows, nullptr, map(0), row0));
    if (!g.resnet) {           // Conv2DNetwork (wavefunctions.py:572-575): act between the convolutions, none behind the last
      for (int l = 1; l < g.n_conv; ++l)
        PROPAGATE(conv(l, rows, map(l - 1), map(l), row0));
      last = map(g.n_conv - 1);
    } else {                   // ResNet2D (wavefunctions.py:766-772; layers.py:226-228): h += second(selu(first(h)))
      for (int l = 1; l + 1 < g.n_conv; l += 2) {
        const float* h = tape ? map(l - 1) : c->cg_fm[0] + moff;
        float* u = tape ? map(l) : c->cg_fm[1] + moff;
        float* hn = tape ? map(l + 1) : c->cg_fm[0] + moff;
        PROPAGATE(conv(l, rows, h, u, row0));
        if (tape) HIPCHK(c, hipMemcpyAsync(hn, h, (size_t)rows * g.N * Fp * sizeof(float), hipMemcpyDeviceToDevice, cg_s(c)));
        PROPAGATE(conv(l + 1, rows, u, hn, row0));
      }
      last = tape ? map(g.n_conv - 1) : c->cg_fm[0] + moff;
    }
    if (!out) continue;        // (taped forward of the SR matvec: the maps are all that is wanted)
    HIPCHK(c, launch_cgen_rowsum(cg_s(c), last, rows, g.N, g.F, Fp, c->cg_sum));
    const WideOnsite on{nullptr, nullptr, nullptr, nullptr, nullptr};
    HIPCHK(c, launch_wide_out_part(cg_s(c), c->cg_sum, 1, c->cg_zero, rows, rowinfo ? rowinfo : c->rowinfo_id, row0,
                                   c->half_jx, p.logit, c->oact, ratio, out, on));
  }
  return VMC_OK;
}

// ---- gradient machinery of the general path, one block of `rows` chains (first chain `row0` of `configs`) at a time
// buffers of the first gradient call
static int cgen_grad_buffers(vmc_ctx* c) {
  const ConvGeom& g = c->cg;
  if (c->cg_tape) return VMC_OK;
  const int T = g.K * g.KW, n_conv = g.n_conv;
  const long long map_floats = c->cg_rows * g.N * cgen_fp(g);
  const int kmax = T * (n_conv > 1 ? g.F : 1) + 1;                      // rows of the largest weight-gradient product
  HIPCHK(c, dalloc(&c->cg_tape, (long long)n_conv * map_floats));
  HIPCHK(c, dalloc(&c->cg_gl, (long long)n_conv * map_floats));
  HIPCHK(c, dalloc(&c->cg_g[0], map_floats));
  HIPCHK(c, dalloc(&c->cg_wpos, c->cg_rows * g.N));
  if (n_conv > 1) HIPCHK(c, dalloc(&c->cg_wt, cgen_off_wt(g, n_conv)));
  c->cg_ws_floats = (long long)CGEN_SPLITK * 2 * kmax * g.F;
  HIPCHK(c, dalloc(&c->cg_ws, c->cg_ws_floats));
  return VMC_OK;
}
static float* cgen_tape(vmc_ctx* c, int l) { return c->cg_tape + (long long)l * c->cg_rows * c->cg.N * cgen_fp(c->cg); }
static float* cgen_gl(vmc_ctx* c, int l) { return c->cg_gl + (long long)l * c->cg_rows * c->cg.N * cgen_fp(c->cg); }

// the input of convolution l gathered into cg_A, as its forward did (from the tape of this block)
static int cgen_gather_input(vmc_ctx* c, int l, int rows, long long row0, const float* configs) {
  const ConvGeom& g = c->cg;
  PROPAGATE(cgen_need_A(c));
  CgenIm2colArgs a;
  memset(&a, 0, sizeof(a));
  a.g = g; a.layer = l; a.Fp = cgen_fp(g); a.rows = rows; a.lda = plan_cgen_lda(g); a.A = c->cg_A; a.pre_act = -1;
  if (l == 0) { a.src = configs; a.row0 = row0; a.bonds = c->bonds ? c->bonds : c->bond_dummy; }
  else { a.src = cgen_tape(c, l - 1); a.pre_act = cgen_in_pre(c, l); }          // (residual blocks: h / the stored selu(u))
  HIPCHK(c, launch_cgen_im2col(c->stream, a));
  return VMC_OK;
}

// dst (+)= the transposed convolution l (>= 1) of G: the inverse gather against the transposed weight image
static int cgen_input_grad(vmc_ctx* c, int l, int rows, const float* G, float* dst, bool accumulate) {
  const ConvGeom& g = c->cg;
  PROPAGATE(cgen_need_A(c));
  CgenIm2colArgs a;
  memset(&a, 0, sizeof(a));
  a.g = g; a.layer = l; a.Fp = cgen_fp(g); a.rows = rows; a.lda = plan_cgen_lda(g); a.A = c->cg_A; a.pre_act = -1;
  a.inverse = 1; a.src = G;
  HIPCHK(c, launch_cgen_im2col(c->stream, a));
  GemmArgs m; memset(&m, 0, sizeof(m));
  m.A = c->cg_A; m.sam = a.lda; m.sak = 1;
  m.B = c->cg_wt + cgen_off_wt(g, l); m.sbk = g.F; m.sbn = 1;
  m.M = rows * g.N; m.N = g.F; m.K = g.K * g.KW * g.F; m.C = dst; m.ldc = a.Fp;
  m.epilogue = accumulate ? 3 : 0; m.splitk = 1;
  HIPCHK(c, launch_gemm(c->stream, m));
  return VMC_OK;
}

// cg_gl[l] = d logit / d z_l for every convolution of the block (the tape of the block in cg_tape):
//   G_{l-1} = (transposed convolution l of G_l) (.) f'(z_{l-1}); residual blocks accumulate both branches into d / d h
static int cgen_backward(vmc_ctx* c, int rows, long long row0, const float* oscale) {
  const ConvGeom& g = c->cg;
  const int Fp = cgen_fp(g), n_conv = g.n_conv;
  const long long M = (long long)rows * g.N;
  float* D = c->cg_g[0];
  HIPCHK(c, launch_cgen_fill(c->stream, cgen_gl(c, n_conv - 1), oscale, row0, rows, g.N, g.F, Fp));
  if (!g.resnet) {
    for (int l = n_conv - 1; l >= 1; --l) {
      PROPAGATE(cgen_input_grad(c, l, rows, cgen_gl(c, l), D, false));
      HIPCHK(c, launch_cgen_dact(c->stream, D, cgen_tape(c, l - 1), g.hact, cgen_post(c), M * Fp, g.F, Fp, cgen_gl(c, l - 1)));
    }
  } else {                     // gl[even l] = d / d h behind block (l / 2): the gradient of the block's second convolution
    for (int l2 = n_conv - 1; l2 >= 2; l2 -= 2) {
      const int l1 = l2 - 1;
      PROPAGATE(cgen_input_grad(c, l2, rows, cgen_gl(c, l2), D, false));                                   // d / d selu(u)
      HIPCHK(c, launch_cgen_dact(c->stream, D, cgen_tape(c, l1), CGEN_PRE_SELU, true, M * Fp, g.F, Fp, cgen_gl(c, l1)));   // d / d u (from the stored selu(u))
      HIPCHK(c, hipMemcpyAsync(cgen_gl(c, l2 - 2), cgen_gl(c, l2), (size_t)M * Fp * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
      PROPAGATE(cgen_input_grad(c, l1, rows, cgen_gl(c, l1), cgen_gl(c, l2 - 2), true));                   // d / d h += through the block
    }
  }
  return VMC_OK;
}

// [w_l ; b_l] sums of convolution l: C1 += [im2col(x_l) | 1]^T G (C1 != nullptr), C2 += [im2col(x_l) | 1]^T (kscale (.) G);
// kscale = per-position weights (cg_wpos).  ONE product (k_gemm: dual, implicit ones row, split-K over the positions)
static int cgen_weight_sums(vmc_ctx* c, int l, int rows, long long row0, const float* configs, const float* G,
                            float* C1, float* C2) {
  const ConvGeom& g = c->cg;
  PROPAGATE(cgen_gather_input(c, l, rows, row0, configs));
  const long long M = (long long)rows * g.N;
  GemmArgs m; memset(&m, 0, sizeof(m));
  m.A = c->cg_A; m.sam = 1; m.sak = plan_cgen_lda(g);                       // A(i, k = position) = im2col[k][i]
  m.B = G; m.sbk = cgen_fp(g); m.sbn = 1;
  m.kscale = c->cg_wpos; m.ones_row = 1;
  m.M = cgen_kdim(g, l) + 1; m.N = g.F; m.K = (int)M; m.ldc = g.F;
  if (C1) { m.dual = 1; m.C = C1 + cgen_off_w(g, l); m.C2 = C2 + cgen_off_w(g, l); }
  else { m.dual = 0; m.C = C2 + cgen_off_w(g, l); }                         // the scaled product alone
  m.epilogue = 3; m.splitk = M >= 4096 ? CGEN_SPLITK : 1; m.workspace = c->cg_ws;
  HIPCHK(c, launch_gemm(c->stream, m));
  return VMC_OK;
}

// Gradient sums of the general path: g1 += sum_b O_b, g2 += sum_b w_b O_b (training.py:545-547), a block of chains at a
// time: taped forward (the map of every convolution), d logit / d z_l of every convolution, then
// d / d W_l = im2col(x_l)^T G_l with the bias as an implicit row of ones, one product per convolution for both sums.
int cgen_gradient_sums(vmc_ctx* c, const float* w) {
  c->cg_sr_tape_rows = 0;                  // (this path overwrites the tapes an SR solve may have left)
  const ConvGeom& g = c->cg;
  ParamSet& p = c->ps[0];
  PROPAGATE(cgen_grad_buffers(c));
  const long long map_floats = c->cg_rows * g.N * cgen_fp(g);
  for (int l = 1; l < g.n_conv; ++l)
    HIPCHK(c, launch_cgen_pack_t(c->stream, p.theta + cgen_off_w(g, l), g.K * g.KW, g.F, c->cg_wt + cgen_off_wt(g, l)));
  if (c->oact != VMC_ACT_EXP_) {
    PROPAGATE(ensure_cache(c, VMC_PSI));
    HIPCHK(c, launch_out_scale(c->stream, p.logit, c->oscale, c->B, c->oact));
  }
  for (long long row0 = 0; row0 < c->B; row0 += c->cg_rows) {
    const int rows = (int)(c->B - row0 < c->cg_rows ? c->B - row0 : c->cg_rows);
    PROPAGATE(cgen_forward(c, VMC_PSI, c->configs, nullptr, rows, nullptr, nullptr, false, c->cg_lnew, c->cg_tape,
                           map_floats, row0));     // (its logits land in cg_lnew[row0 ..]: unused)
    HIPCHK(c, launch_cgen_wpos(c->stream, w, row0, rows, g.N, c->cg_wpos));
    PROPAGATE(cgen_backward(c, rows, row0, c->oact != VMC_ACT_EXP_ ? c->oscale : nullptr));
    for (int l = g.n_conv - 1; l >= 0; --l)
      PROPAGATE(cgen_weight_sums(c, l, rows, row0, c->configs, cgen_gl(c, l), c->acc, c->acc + c->P));
  }
  return VMC_OK;
}

// SR matvec of the general path over the `n_rows` stored chains (sr_cfg): u = sum_b (t_b - c) O_b with t_b = O_b . v,
// u[P] = sum_b (t_b - c).  k_sr_q forms q = u / n - <O> u[P] / n + lambda p, which is S v + lambda v for ANY constant c
// subtracted from every t_b -- and with c = the mean of t the cancellation of <O (O.v)> - <O><O.v> happens per sample,
// before the fp32 sums (the uncentred form of this matvec met the 5e-4 bound on S v but its solutions missed the 1 %
// bound on O_c x).  The constant must be the same on every rank: a sharded solve all-reduces sum_b t_b between the two
// phases (sr_solve_impl), a single rank takes its own mean.  Only the chains are stored: every CG iteration re-runs the
// taped forward and the backward of a block; t_b = sum_l < G_l , im2col(x_l) V_l + v_l > is one more product per
// convolution against the slice of v (phase 1), then the weight sums with t_b - c as the k-scale (phase 2; several
// blocks: a second forward / backward pass, the mean needs every t first).
// (round 6: when the stored chains are ONE block, its tapes and G_l stay valid for every CG iteration of the running solve
// -- the parameters do not change inside a solve; vmc_sr_begin and the gradient path, which shares the buffers, reset the mark)
static int cgen_sr_fwd_bwd(vmc_ctx* c, long long row0, int rows, long long n_rows) {
  const char* keep_env = getenv("CGS_VMC_SR_KEEP_TAPE");                 // 0: recompute every iteration (A/B; read per call)
  const bool one_block = row0 == 0 && rows == n_rows && !(keep_env && atoi(keep_env) == 0);
  if (one_block && c->cg_sr_tape_rows == n_rows) return VMC_OK;
  c->cg_sr_tape_rows = 0;
  const long long map_floats = c->cg_rows * c->cg.N * cgen_fp(c->cg);
  PROPAGATE(cgen_forward(c, VMC_PSI, c->sr_cfg, nullptr, rows, nullptr, nullptr, false, nullptr, c->cg_tape, map_floats, row0));
  PROPAGATE(cgen_backward(c, rows, row0, nullptr));
  if (one_block) c->cg_sr_tape_rows = n_rows;
  return VMC_OK;
}
int cgen_sr_phase1(vmc_ctx* c, const float* v, int n_rows) {        // sr_t[b] = O_b . v
  const ConvGeom& g = c->cg;
  ParamSet& p = c->ps[0];
  PROPAGATE(cgen_grad_buffers(c));
  if (!c->cg_td) { HIPCHK(c, dalloc(&c->cg_td, c->cg_rows)); HIPCHK(c, dalloc(&c->cg_centre, 1)); }
  const int Fp = cgen_fp(g), lda = plan_cgen_lda(g);
  for (int l = 1; l < g.n_conv; ++l)
    HIPCHK(c, launch_cgen_pack_t(c->stream, p.theta + cgen_off_w(g, l), g.K * g.KW, g.F, c->cg_wt + cgen_off_wt(g, l)));
  for (long long row0 = 0; row0 < n_rows; row0 += c->cg_rows) {
    const int rows = (int)(n_rows - row0 < c->cg_rows ? n_rows - row0 : c->cg_rows);
    PROPAGATE(cgen_sr_fwd_bwd(c, row0, rows, n_rows));
    for (int l = 0; l < g.n_conv; ++l) {
      PROPAGATE(cgen_gather_input(c, l, rows, row0, c->sr_cfg));
      GemmArgs m; memset(&m, 0, sizeof(m));
      m.A = c->cg_A; m.sam = lda; m.sak = 1;
      m.B = v + cgen_off_w(g, l); m.sbk = g.F; m.sbn = 1;
      m.M = rows * g.N; m.N = g.F; m.K = cgen_kdim(g, l); m.C = c->cg_g[0]; m.ldc = Fp;
      m.bias = v + cgen_off_b(g, l); m.epilogue = 4; m.splitk = 1;
      HIPCHK(c, launch_gemm(c->stream, m));
      HIPCHK(c, launch_cgen_pairdot(c->stream, c->cg_g[0], cgen_gl(c, l), rows, g.N, g.F, Fp, c->cg_td, l == 0));
    }
    HIPCHK(c, launch_cgen_tstore(c->stream, c->cg_td, rows, c->sr_t + row0));
  }
  return VMC_OK;
}
int cgen_sr_phase2(vmc_ctx* c, int n_rows) {                         // sr_u[0 .. P) += sum_b (t_b - *cg_centre) O_b
  const ConvGeom& g = c->cg;
  const bool one_block = n_rows <= c->cg_rows;       // (then the tapes and G_l of phase 1 are still in place)
  for (long long row0 = 0; row0 < n_rows; row0 += c->cg_rows) {
    const int rows = (int)(n_rows - row0 < c->cg_rows ? n_rows - row0 : c->cg_rows);
    if (!one_block) PROPAGATE(cgen_sr_fwd_bwd(c, row0, rows, n_rows));
    HIPCHK(c, launch_cgen_wpos_centred(c->stream, c->sr_t, c->cg_centre, row0, rows, g.N, c->cg_wpos));
    for (int l = g.n_conv - 1; l >= 0; --l)
      PROPAGATE(cgen_weight_sums(c, l, rows, row0, c->sr_cfg, cgen_gl(c, l), nullptr, c->sr_u));
  }
  return VMC_OK;
}
// one rank
int cgen_sr_matvec(vmc_ctx* c, const float* v, int n_rows) {
  PROPAGATE(cgen_sr_phase1(c, v, n_rows));
  HIPCHK(c, launch_cgen_tmean(c->stream, c->sr_t, n_rows, c->cg_centre, c->sr_u + c->P));
  return cgen_sr_phase2(c, n_rows);
}


}  // namespace vmcapi

