// k_tail16s -- EXPERIMENT (CGS_VMC_SPLIT_BF16=1): the row kernel of the 256-unit
// relu fully-connected ansatz with fp32 results from the BF16 matrix cores.
//
// gfx950 multiplies f32 x f32 on the matrix pipe at 1/16 of the bf16 rate (MI355X_MICROARCH.md).  Every
// f32 operand is split into three bf16 terms, x = hi + mid + lo (hi = bf16(x), mid = bf16(x - hi),
// lo = bf16(x - hi - mid): 24 bits of significand between them), and a product w x is the six bf16
// products whose weight is above 2^-24 of the leading one -- hi hi, hi mid, mid hi, hi lo, lo hi, mid mid
// -- accumulated in fp32 by v_mfma_f32_16x16x32_bf16: 6 / 16 of the native fp32 matrix time.  The
// weights are split once per parameter change (k_pack_split), the activations in the layer epilogue
// that touches them anyway (11 VALU instructions per pair of values).
//
// Same transposed, register-resident scheme as k_tail16 (tail16.hpp): a wave owns 32 rows as two 16-row
// halves and all 256 units; the accumulator of output tile `to` (lane (row j, g): units 16 to + 4 g + r)
// feeds the next layer with no lane movement because the k slots of a 32-deep bf16 MFMA step are ASSIGNED
// to units accordingly: slot 8 g + s of k-step kt is unit 16 (2 kt + (s >> 2)) + 4 g + (s & 3), i.e. lane
// (j, g) fills its eight slots from its own registers of tiles 2 kt and 2 kt + 1; the weight image is
// packed in that slot order.  First layer as in k_tail16: cached z1 + the rank-2 exchange update, fp32.
// fully_connected, relu, Hp = 256 only; the headline benchmark stays on the native fp32 kernel.
#include "common.hpp"

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define SPLIT_NT 16
#define SPLIT_KT 8                       // 32-deep k-steps per layer
#define SPLIT_ITEMS (SPLIT_NT * SPLIT_KT)

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {   // lo 16 bits = bf16(a), hi = bf16(b); round to nearest even
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// two f32 -> their three packed bf16 terms
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = cvt_pk_bf16(r0, r1);
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
  l = cvt_pk_bf16(s0, s1);
}

// the eight k slots of a lane for k-step kt: registers of unit tiles 2 kt and 2 kt + 1
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, u32x4& H, u32x4& M, u32x4& L) {
  unsigned h[4], m[4], l[4];
  split2(a[0], a[1], h[0], m[0], l[0]);
  split2(a[2], a[3], h[1], m[1], l[1]);
  split2(b[0], b[1], h[2], m[2], l[2]);
  split2(b[2], b[3], h[3], m[3], l[3]);
  H = u32x4{h[0], h[1], h[2], h[3]};
  M = u32x4{m[0], m[1], m[2], m[3]};
  L = u32x4{l[0], l[1], l[2], l[3]};
}

__device__ __forceinline__ f32x4 mfma_bf16(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// theta -> split weight image: [layer][to][kt][term h, m, l][64 lanes][4 dwords]; lane (i, g), dword d,
// half e: W_l[unit_in(kt, 8 g + 2 d + e)][16 to + i] with unit_in(kt, s') = 16 (2 kt + (s' >> 2 & 1)) + 4 g + (s' & 3)
// for slot s' = 2 d + e within the lane's eight
__global__ void k_pack_split(const float* __restrict__ theta, int H, ParamLayout lay, unsigned* __restrict__ out) {
  const long long per_layer = (long long)SPLIT_ITEMS * 3 * 256;
  const long long total = (long long)lay.n_hh * SPLIT_ITEMS * 256;        // one thread per (layer, item, lane, dword)
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int d = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    const long long it = idx >> 8;
    const int item = (int)(it % SPLIT_ITEMS), l = (int)(it / SPLIT_ITEMS);
    const int to = item / SPLIT_KT, kt = item % SPLIT_KT, i = lane & 15, g = lane >> 4;
    const float* W = theta + lay.off_h0 + (long long)l * ((long long)H * H + H);   // [in][out] row-major (plan_off_w)
    float x[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int s = 2 * d + e;
      const int uin = 16 * (2 * kt + (s >> 2)) + 4 * g + (s & 3), uout = 16 * to + i;
      x[e] = (uin < H && uout < H) ? W[(long long)uin * H + uout] : 0.f;
    }
    unsigned h, m, lo;
    split2(x[0], x[1], h, m, lo);
    unsigned* o = out + l * per_layer + ((long long)item * 3) * 256 + lane * 4 + d;
    o[0] = h; o[256] = m; o[512] = lo;
  }
}

hipError_t launch_pack_split(hipStream_t s, const float* theta, int H, const ParamLayout& lay, unsigned* out) {
  if (lay.n_hh <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_pack_split, dim3(256), dim3(256), 0, s, theta, H, lay, out);
  return hipGetLastError();
}

long long pack_split_dwords(int n_hh) { return (long long)(n_hh > 0 ? n_hh : 1) * SPLIT_ITEMS * 3 * 256; }

// Weights stream L2 -> registers per wave through a ring of SPLIT_RD items (item = the three 1 KiB
// fragments hi, mid, lo of one (output tile, k-step)), issued RD - 1 items ahead across layer and tile
// boundaries, every issue unconditional -- k_tail16's scheme.  At full matrix speed the four waves of a
// CU would pull 4 x 3 KiB per 192 cycles = 64 B/clk, the whole L1 bandwidth: this kernel is bound by
// its weight stream, not by the matrix pipe.  (Sharing the stream through LDS -- one 24 KiB stage per
// output tile fetched once per workgroup, three buffers, a barrier per stage -- was measured SLOWER:
// 0.81 ms against 0.65 ms; the fragments still have to travel LDS -> registers, and the barrier exposes
// every wave's LDS latency once per 1,536 cycles.)
// (a six-deep ring does not divide the 128 items of a layer -- its slots would not continue across layers --
// and timed the same as four: 0.64 ms; eight spills)
#define SPLIT_RD 4

template <bool RATIO>
__global__ __launch_bounds__(256) void k_tail16s(TailArgs a, const unsigned* __restrict__ p16s) {
  constexpr int NT = SPLIT_NT, Hp = 256, KT = SPLIT_KT, NI = SPLIT_ITEMS, RD = SPLIT_RD;
  static_assert(NI % RD == 0, "ring slots continue across layers only if RD divides the items per layer");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, j = lane & 15;
  const int n_rows = a.n_rows_dev ? *a.n_rows_dev : a.n_rows;
  const int n_hidden = a.n_hidden;
  const float bout = a.pp.bout[0];
  const int oact = a.oact;

  typedef const __attribute__((address_space(1))) u32x4* gu4_p;
  struct Frag { u32x4 h, m, l; };
  Frag ring[RD];
  const unsigned* wbase = p16s;
  auto issue = [&](int l, int item) {
    gu4_p p = (gu4_p)(wbase + ((long long)l * NI + item) * 3 * 256) + lane;
    Frag f;
    f.h = p[0]; f.m = p[64]; f.l = p[128];
    return f;
  };
#pragma unroll
  for (int i = 0; i < RD - 1; ++i) ring[i] = issue(0, i);

  struct Desc { const float* zb; const float* wa; const float* wb; float coef, lbase, hjx; int row, valid; };
  auto describe = [&](int tile, int half) {
    Desc d;
    d.row = tile * 128 + wave * 32 + 16 * half + j;
    d.valid = d.row < n_rows;
    const int2 ri = a.rowinfo[d.valid ? d.row : n_rows - 1];   // {chain, +-(bond+1) or 0}
    const int bs = ri.y;
    const int bond = (bs > 0 ? bs : -bs) - (bs != 0 ? 1 : 0);
    d.coef = bs > 0 ? -2.f : (bs < 0 ? 2.f : 0.f);
    const int2 ab = a.bonds[bond];
    d.wa = a.pp.w1p + (long long)ab.x * Hp;
    d.wb = a.pp.w1p + (long long)ab.y * Hp;
    d.zb = a.z1 + (long long)ri.x * Hp;
    d.lbase = RATIO ? a.logit_base[ri.x] : 0.f;
    d.hjx = RATIO ? a.half_jx[bond] : 0.f;
    return d;
  };
  auto gather = [&](const Desc& d, int t) {      // first-layer activations of unit tile t: relu(z1 + coef (W1[i] - W1[j]))
    const int off = 16 * t + 4 * g;
    const f32x4 z = *(const f32x4*)(d.zb + off), x = *(const f32x4*)(d.wa + off), y = *(const f32x4*)(d.wb + off);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = vmc_act<VMC_ACT_RELU_>(fmaf(d.coef, x[e] - y[e], z[e]));
    return v;
  };

  int tile = blockIdx.x;
  if (tile * 128 + wave * 32 >= n_rows) return;   // wave-uniform; no barriers in this kernel
  Desc cur[2], nxt[2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) cur[hf] = describe(tile, hf);
  u32x4 Xh[2][KT], Xm[2][KT], Xl[2][KT];         // B operands of the current layer: [half][k-step] x (hi, mid, lo)
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const f32x4 v0 = gather(cur[hf], 2 * kt), v1 = gather(cur[hf], 2 * kt + 1);
      split8(v0, v1, Xh[hf][kt], Xm[hf][kt], Xl[hf][kt]);
    }

  for (;;) {
    const int next_tile = tile + gridDim.x;
    const bool has_next = next_tile * 128 + wave * 32 < n_rows;   // wave-uniform
    int opaque0 = 0;
    asm volatile("" : "+s"(opaque0));   // keeps tile-invariant bias / w_out loads inside the loop
    asm volatile("" : "+s"(wbase));     // ... and the per-item weight addresses
    const int nt_safe = has_next ? next_tile : tile;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) nxt[hf] = describe(nt_safe, hf);
    float part[2] = {0.f, 0.f};
    // activations of the layer being produced; under the LAST layer (whose outputs go straight into the
    // output dot) the same registers collect the first-layer activations of the NEXT row tile -- no LDS
    f32x4 out[2][NT];
    for (int l = 0; l < n_hidden; ++l) {
      const bool last = l + 1 == n_hidden;
      const float* __restrict__ bl = a.pp.bh + l * Hp + opaque0;
      const float* __restrict__ wop = a.pp.woutp + opaque0;
#pragma unroll
      for (int to = 0; to < NT; ++to) {
        const f32x4 bias = *(const f32x4*)(bl + 16 * to + 4 * g);
        f32x4 acc0 = bias, acc1 = bias;
        // unit tile `to` of the NEXT row tile is gathered under this output tile (only the last layer's copy is kept)
        const f32x4 gv0 = gather(nxt[0], to), gv1 = gather(nxt[1], to);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const int item = to * KT + kt, nx = item + RD - 1;
          if (nx < NI) ring[nx % RD] = issue(l, nx);
          else ring[nx % RD] = issue(last ? 0 : l + 1, nx - NI);
          __builtin_amdgcn_sched_barrier(0);
          const Frag w = ring[item % RD];
          // smallest terms first
          acc0 = mfma_bf16(w.l, Xh[0][kt], acc0); acc1 = mfma_bf16(w.l, Xh[1][kt], acc1);
          acc0 = mfma_bf16(w.h, Xl[0][kt], acc0); acc1 = mfma_bf16(w.h, Xl[1][kt], acc1);
          acc0 = mfma_bf16(w.m, Xm[0][kt], acc0); acc1 = mfma_bf16(w.m, Xm[1][kt], acc1);
          acc0 = mfma_bf16(w.m, Xh[0][kt], acc0); acc1 = mfma_bf16(w.m, Xh[1][kt], acc1);
          acc0 = mfma_bf16(w.h, Xm[0][kt], acc0); acc1 = mfma_bf16(w.h, Xm[1][kt], acc1);
          acc0 = mfma_bf16(w.h, Xh[0][kt], acc0); acc1 = mfma_bf16(w.h, Xh[1][kt], acc1);
        }
        if (last) {
          out[0][to] = gv0;
          out[1][to] = gv1;
          const f32x4 wo = *(const f32x4*)(wop + 16 * to + 4 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            part[0] = fmaf(vmc_act<VMC_ACT_RELU_>(acc0[e]), wo[e], part[0]);
            part[1] = fmaf(vmc_act<VMC_ACT_RELU_>(acc1[e]), wo[e], part[1]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) { acc0[e] = vmc_act<VMC_ACT_RELU_>(acc0[e]); acc1[e] = vmc_act<VMC_ACT_RELU_>(acc1[e]); }
          out[0][to] = acc0; out[1][to] = acc1;
        }
      }
      if (!last) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) split8(out[hf][2 * kt], out[hf][2 * kt + 1], Xh[hf][kt], Xm[hf][kt], Xl[hf][kt]);
      }
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      float p = part[hf];
      p += __shfl_xor(p, 16);
      p += __shfl_xor(p, 32);
      const float logit = p + bout;
      if (cur[hf].valid && g == 0) {
        if (RATIO) a.out[cur[hf].row] = cur[hf].hjx * vmc_out_ratio(oact, logit, cur[hf].lbase);
        else a.out[cur[hf].row] = logit;
      }
    }
    if (!has_next) break;
    tile = next_tile;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) cur[hf] = nxt[hf];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) split8(out[hf][2 * kt], out[hf][2 * kt + 1], Xh[hf][kt], Xm[hf][kt], Xl[hf][kt]);
  }
}

hipError_t launch_tail16_split(hipStream_t s, const TailArgs& a, const unsigned* p16s, bool ratio_mode) {
  if (a.n_rows <= 0) return hipSuccess;
  const int tiles = (a.n_rows + 127) / 128;
  const int persistent = a.num_cus > 0 ? a.num_cus : 256;
  const dim3 grid(tiles < persistent ? tiles : persistent), block(256);
  const size_t lds = 0;
  hipError_t e;
  if (ratio_mode) {
    e = hipFuncSetAttribute((const void*)k_tail16s<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_tail16s<true>), grid, block, lds, s, a, p16s);
  } else {
    e = hipFuncSetAttribute((const void*)k_tail16s<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_tail16s<false>), grid, block, lds, s, a, p16s);
  }
  return hipGetLastError();
}
