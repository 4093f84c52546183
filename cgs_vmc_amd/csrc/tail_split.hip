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
#include <cstdlib>

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

// ---------------------------------------------------------------------------------------------------------
// k_tail16r (round 5): the same arithmetic, the same order of additions -- results are the bits of k_tail16s --
// with the weight stream fetched ONCE PER WORKGROUP instead of once per wave.  k_tail16s pulls 4 x 3 KiB per
// 192 matrix cycles through the CU's L1 (64 B/clk, all of it: MFMA busy 0.417, SQ_WAIT_ANY 37 % of the wave
// cycles, profiles/r4_split_pmc_summary.txt).  Here the three-term fragments of one output tile (8 k-steps x
// 3 KiB = one 24 KiB STAGE) travel L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip), each
// of the four waves issuing a quarter of a stage (six 1 KiB pieces), into a ring of SPLIT_RING stages; a wave
// reads its A operands with three ds_read_b128 per item, one item ahead of its MFMAs.
//   Protocol per stage st (every wave runs the same stage sequence; ONE s_barrier per stage = 1,536 matrix
//   cycles, and nothing waits at it in the steady state -- the stage it certifies is the NEXT one):
//     top of st:  s_waitcnt vmcnt(6)   my pieces of stage st+1 have landed (only st+2's six may be in flight)
//                 s_barrier            => all of st+1 has landed; every wave has finished READING st-1
//                 issue my six pieces of stage st+3 into the slot of st-1
//     items 0..7: fragments of item i+1 are read (across the st -> st+1 boundary too: st+1 is certified)
//                 while the twelve MFMAs of item i run
//   A wave's reads of stage st are complete before it reaches the next barrier (its MFMAs consumed them), so
//   the DMA into a freed slot never races a read.  Every wave drains its DMA before it exits.
// The LDS reads are inline asm (LDS_STEP): the compiler never sees an LDS access that a DMA could alias.
#define SPLIT_RING 4
// DIAGNOSTIC builds only (-DVMC_SPLIT_ABLATE=mask, tools/split_ablate.sh; results are garbage, only the time is read):
// 1 no operand split, 2 no gather of the next row tile, 4 no DMA issue, 8 no barrier, 16 no MFMAs, 32 no LDS reads
#ifndef VMC_SPLIT_ABLATE
#define VMC_SPLIT_ABLATE 0
#endif
#define SPLIT_STAGE_BYTES (SPLIT_KT * 3 * 1024)

// EVERY vector-memory operation of the stage loop is inline asm with hand-counted waits: the six DMA pieces, the
// six gather loads of the next row tile, and (from LDS) bias and w_out.  Mixed with compiler-visible loads the
// compiler's vmcnt waits come out wrong for this pipeline either way: as a builtin the DMA makes it wait vmcnt(0)
// for the gathers (the pieces just issued must land mid-stage), as asm it is invisible to the compiler's count and
// a wait for a load issued a stage earlier forces the loads just issued to land.
// ONE 1 KiB piece of a stage: global [stage + voff + OFF] -> LDS [lds + OFF + 16 lane]; the instruction offset of an
// LDS-DMA load is added to the global address AND to the LDS address.  M0 is saved and restored.
#define DMA1(STAGE, VOFF, LDS, OFF)                                                                              \
  do {                                                                                                           \
    unsigned sv_;                                                                                                \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3 offset:%4\n\t" \
                 "s_mov_b32 m0, %0"                                                                              \
                 : "=&s"(sv_) : "s"(LDS), "v"(VOFF), "s"(STAGE), "n"(OFF) : "memory");                           \
  } while (0)
// the three 1 KiB pieces of one row half of the NEXT row tile (z1 of the chain, the two W1 rows of the bond), by
// LDS-DMA with per-lane addresses into this wave's staging area: lane L fetches 16 bytes at its pointer + OFF and
// they land at LDS [base + k KiB + 16 L].  The instruction offset moves BOTH addresses, so M0 is set OFF lower.
// (No register ever receives data behind the compiler's back here: an asynchronous load into registers that the
// compiler believes already valid -- the first form of this gather -- is at the mercy of its live-range splitting:
// it copied the "value" elsewhere before it had landed, the landing then overwrote whatever had moved in: a
// pointer, a memory fault.)
#define GATHER_DMA3(PZ, PX, PY, LDS, OFF)                                                                          \
  do {                                                                                                             \
    unsigned sv_;                                                                                                  \
    asm volatile("s_mov_b32 %0, m0\n\ts_sub_u32 m0, %1, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off offset:%5\n\t" \
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off offset:%5\n\t"              \
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off offset:%5\n\t"              \
                 "s_mov_b32 m0, %0"                                                                                \
                 : "=&s"(sv_) : "s"(LDS), "v"(PZ), "v"(PX), "v"(PY), "n"(OFF) : "memory", "scc");                  \
  } while (0)

// A-operand fragments from the ring, as inline asm (the compiler must not see an LDS access next to the DMA, and
// its scheduler, short of registers, sinks ordinary LDS loads next to their use).  ONE statement per item: wait
// for the reads of the item about to be multiplied (set s, issued by the previous statement), then issue the
// three reads of the item behind it into the OTHER set t.  Both sets are in/out operands: the MFMAs of item i
// read set s = outputs of statement i (they follow it), and statement i + 1 redefines set s (it follows them) --
// so the scheduler keeps [statement i, MFMAs of item i, statement i + 1] in that order and the reads of item
// i + 1 are in flight under the twelve MFMAs of item i.
#define LDS_STEP(S, T, ADDR, OFF)                                                                          \
  asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %3, %6 offset:%7\n\tds_read_b128 %4, %6 offset:%7+1024\n\t" \
               "ds_read_b128 %5, %6 offset:%7+2048"                                                         \
               : "+v"(wh[S]), "+v"(wm[S]), "+v"(wl[S]), "+v"(wh[T]), "+v"(wm[T]), "+v"(wl[T])              \
               : "v"(ADDR), "n"(OFF))

template <bool B> struct bool_c { static constexpr bool value = B; };

// Where the time went before this shape (tools/split_ablate.sh, profiles/r5_split_ablate.txt; 0.562 ms with all six
// pieces and six gather loads issued back to back at the top of a stage): the gather of the next row tile 0.124 ms --
// lane (row j, k group g) = 16 g + j loading its own 16 bytes puts four ROWS into every quad of lanes, 64 tag
// look-ups per load, in every layer although only the last layer's copy is kept; the six DMA pieces 0.087 ms (a wave
// stalls on the address unit's queue when they follow each other); the barrier 0.044 ms; the operand split 0.019 ms;
// the LDS reads 0.022 ms.  Hence:
//   * one DMA piece per item (items 0 .. 5), so each is issued into an idle address unit under MFMAs;
//   * the gather only in the last layer, in LOAD layout -- lane L takes row L >> 2, 16-byte piece L & 3: a quad reads
//     64 contiguous bytes, 16 look-ups per load -- one row half at a time (items 0 and 3), by LDS-DMA into a 6 KiB
//     staging area of the wave; lane (j, g) reads piece 4 j + g back (the transpose is the read address) and does
//     the first-layer arithmetic in the MFMA operand layout;
//   * the epilogue of output tile `to` (relu / output dot) written into stage to + 1 behind its second item, where
//     it fills MFMA shadows instead of standing between two stages;
//   * the layer loop body twice, with `last` a compile-time constant: no branch inside a layer's sixteen stages.
template <bool RATIO>
__global__ __launch_bounds__(256) void k_tail16r(TailArgs a, const unsigned* __restrict__ p16s) {
  constexpr int NT = SPLIT_NT, Hp = 256, KT = SPLIT_KT, R = SPLIT_RING;
  extern __shared__ __attribute__((aligned(16))) char s_ring[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, j = lane & 15;
  const int n_rows = a.n_rows_dev ? *a.n_rows_dev : a.n_rows;
  const int n_hidden = a.n_hidden;
  const float bout = a.pp.bout[0];
  const int oact = a.oact;
  int tile = blockIdx.x;
  if (tile * 128 >= n_rows || n_hidden <= 0) return;        // workgroup-uniform: every wave of a live workgroup stays

  // issue cursor: the stage (layer il, output tile ito) whose pieces are being fetched, into slot islot
  const unsigned lds0 = (unsigned)(size_t)s_ring;
  const unsigned voff0 = wave * 6144 + lane * 16, voff1 = voff0 + 4096;
  int il = 0, ito = 0, islot = 0;
  const unsigned* istage = p16s;              // global address of the stage under the cursor
  unsigned ilds = lds0 + wave * 6144;         // this wave's part of the cursor's slot
  auto advance = [&]() {
    if (++ito == NT) { ito = 0; if (++il == n_hidden) il = 0; }
    if (++islot == R) islot = 0;
    istage = p16s + (long long)(il * NT + ito) * (KT * 3 * 256);
    ilds = lds0 + islot * SPLIT_STAGE_BYTES + wave * 6144;
  };
  // (the scalar operands pass through readfirstlane: short of SGPRs the compiler keeps uniform values in VGPRs and
  // hands such a register to an "s" operand as it stands -- `s_mov_b32 m0, v1` does not assemble)
  auto sgpr_ptr = [](const unsigned* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const unsigned*)(((unsigned long long)hi << 32) | lo);
  };
#define DMA_PIECE(P)                                                                   \
  do {                                                                                 \
    const unsigned* st_ = sgpr_ptr(istage);                                            \
    const unsigned ld_ = __builtin_amdgcn_readfirstlane(ilds);                         \
    if ((P) < 4) DMA1(st_, voff0, ld_, (P) * 1024);                                    \
    else DMA1(st_, voff1, ld_ + 4096, ((P) - 4) * 1024);                               \
  } while (0)
#pragma unroll
  for (int i = 0; i < R - 1; ++i) {
    DMA_PIECE(0); DMA_PIECE(1); DMA_PIECE(2); DMA_PIECE(3); DMA_PIECE(4); DMA_PIECE(5);
    advance();
  }
  // bias of every H x H layer and w_out behind the ring: [n_hidden][256] + [256] floats
  float* s_bias = (float*)(s_ring + R * SPLIT_STAGE_BYTES);
  for (int l = 0; l < n_hidden; ++l) s_bias[l * Hp + threadIdx.x] = a.pp.bh[l * Hp + threadIdx.x];
  s_bias[n_hidden * Hp + threadIdx.x] = a.pp.woutp[threadIdx.x];

  // rows in the MFMA operand layout (lane (j, g): row j of a half): the output side
  struct Desc { float lbase, hjx, coef; int row, valid; };
  auto describe = [&](int tile, int half) {
    Desc d;
    d.row = tile * 128 + wave * 32 + 16 * half + j;
    d.valid = d.row < n_rows;
    const int2 ri = a.rowinfo[d.valid ? d.row : n_rows - 1];   // {chain, +-(bond+1) or 0}
    const int bs = ri.y;
    const int bond = (bs > 0 ? bs : -bs) - (bs != 0 ? 1 : 0);
    d.coef = bs > 0 ? -2.f : (bs < 0 ? 2.f : 0.f);
    d.lbase = RATIO ? a.logit_base[ri.x] : 0.f;
    d.hjx = RATIO ? a.half_jx[bond] : 0.f;
    return d;
  };
  // rows in the LOAD layout (lane L: row L >> 2 of a half, 16-byte piece L & 3 of a unit tile): the gather side
  struct Src { const float* zb; const float* wa; const float* wb; };
  auto source = [&](int tile, int half) {
    Src d;
    const int row = tile * 128 + wave * 32 + 16 * half + (lane >> 2);
    const int2 ri = a.rowinfo[row < n_rows ? row : n_rows - 1];
    const int bs = ri.y;
    const int bond = (bs > 0 ? bs : -bs) - (bs != 0 ? 1 : 0);
    const int2 ab = a.bonds[bond];
    d.wa = a.pp.w1p + (long long)ab.x * Hp + 4 * (lane & 3);
    d.wb = a.pp.w1p + (long long)ab.y * Hp + 4 * (lane & 3);
    d.zb = a.z1 + (long long)ri.x * Hp + 4 * (lane & 3);
    return d;
  };
  // this wave's gather staging: [2 halves][z, x, y][1 KiB], behind the bias image; lane (j, g) reads piece 4 j + g
  char* s_stage = s_ring + R * SPLIT_STAGE_BYTES + (n_hidden + 1) * 1024 + wave * 6144;
  const unsigned stage_lds = (unsigned)(size_t)s_stage;
  const f32x4* s_mine = (const f32x4*)s_stage + (4 * j + g);
  auto first_layer = [&](float coef, const f32x4& z, const f32x4& x, const f32x4& y) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = vmc_act<VMC_ACT_RELU_>(fmaf(coef, x[e] - y[e], z[e]));
    return v;
  };
  Desc cur[2];
  Src nsrc[2];
  float ncoef[2];
  u32x4 Xh[2][KT], Xm[2][KT], Xl[2][KT];         // B operands of the current layer: [half][k-step] x (hi, mid, lo)
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    cur[hf] = describe(tile, hf);
    // (the first row tile of a workgroup: ordinary loads in the operand layout, once)
    const int2 ri = a.rowinfo[cur[hf].valid ? cur[hf].row : n_rows - 1];
    const int bs = ri.y;
    const int2 ab = a.bonds[(bs > 0 ? bs : -bs) - (bs != 0 ? 1 : 0)];
    const float* zb = a.z1 + (long long)ri.x * Hp + 4 * g;
    const float* wa = a.pp.w1p + (long long)ab.x * Hp + 4 * g;
    const float* wb = a.pp.w1p + (long long)ab.y * Hp + 4 * g;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      f32x4 v[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int off = 16 * (2 * kt + u);
        v[u] = first_layer(cur[hf].coef, *(const f32x4*)(zb + off), *(const f32x4*)(wa + off), *(const f32x4*)(wb + off));
      }
      split8(v[0], v[1], Xh[hf][kt], Xm[hf][kt], Xl[hf][kt]);
    }
  }

  const unsigned ring0 = lds0 + lane * 16;
  int slot = 0;                                  // slot of the stage being multiplied
  // stages 0 and 1 certified here (and the bias image written); from then on the barrier at the top of stage st
  // certifies st+1
  asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  u32x4 wh[2], wm[2], wl[2];                     // fragments of the item being multiplied / the one behind it
  wh[1] = wm[1] = wl[1] = u32x4{0, 0, 0, 0};
  asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:1024\n\tds_read_b128 %2, %3 offset:2048"
               : "=&v"(wh[0]), "=&v"(wm[0]), "=&v"(wl[0]) : "v"(ring0));
  bool first_stage = true;
  f32x4 bias = ((const f32x4*)s_bias)[g];       // of the stage about to be multiplied (layer 0, output tile 0)

  float part[2];
  f32x4 out[2][NT];
  // the sixteen stages of one layer; `last` is a compile-time constant (bool_c)
  auto layer = [&](auto last_c, const int l) __attribute__((always_inline)) {
    constexpr bool last = decltype(last_c)::value;
    const f32x4* s_wo = (const f32x4*)(s_bias + n_hidden * Hp) + g;        // + 4 to: units 16 to + 4 g .. + 3
    const f32x4* s_bl = (const f32x4*)(s_bias + l * Hp) + g;
    const f32x4* s_bnl = (const f32x4*)(s_bias + (last ? 0 : l + 1) * Hp) + g;
    f32x4 pacc0, pacc1, pwo;                     // accumulators and w_out of the previous output tile (deferred epilogue)
    f32x4 gv[2];                                 // first-layer activations of unit tile `to` of the next row tile
    auto epilogue = [&](int t, const f32x4& acc0, const f32x4& acc1, const f32x4& wo) {
      if (last) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          part[0] = fmaf(vmc_act<VMC_ACT_RELU_>(acc0[e]), wo[e], part[0]);
          part[1] = fmaf(vmc_act<VMC_ACT_RELU_>(acc1[e]), wo[e], part[1]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { out[0][t][e] = vmc_act<VMC_ACT_RELU_>(acc0[e]); out[1][t][e] = vmc_act<VMC_ACT_RELU_>(acc1[e]); }
      }
    };
#pragma unroll
    for (int to = 0; to < NT; ++to) {
      // ---- top of a stage
      if (!first_stage) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // only this wave's six pieces of stage st+2 may be in flight
        if (!(VMC_SPLIT_ABLATE & 8)) __builtin_amdgcn_s_barrier();
      }
      first_stage = false;
      // w_out of this tile (epilogue) and the bias of the NEXT stage (its top), from LDS; the lgkmcnt(0) of the
      // next LDS_STEP covers them, the empty asm behind it carries the values past it
      // w_out of this tile (epilogue) and the bias of the NEXT stage (its top): ordinary loads of the LDS image that
      // was written once before the first barrier (nothing to order them against; the compiler counts their waits)
      const f32x4 wo = s_wo[4 * to];
      const f32x4 bias_next = to + 1 < NT ? s_bl[4 * (to + 1)] : s_bnl[0];
      const unsigned sbase = ring0 + slot * SPLIT_STAGE_BYTES;
      const int nslot = slot + 1 == R ? 0 : slot + 1;
      const unsigned nbase = ring0 + nslot * SPLIT_STAGE_BYTES;
      f32x4 acc0 = bias, acc1 = bias;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        if (VMC_SPLIT_ABLATE & 32) asm volatile("" : "+v"(wh[0]), "+v"(wm[0]), "+v"(wl[0]), "+v"(wh[1]), "+v"(wm[1]), "+v"(wl[1]));
        else if (kt + 1 < KT) LDS_STEP(kt & 1, (kt + 1) & 1, sbase, (kt + 1) * 3072);
        else LDS_STEP(kt & 1, (kt + 1) & 1, nbase, 0);
        // vector-memory operations of this item, in this order: the gather of one row half (last layer: half 0 in item
        // 0, consumed in item 3; half 1 in item 3 into the same registers, consumed in item 6), then one DMA piece of
        // stage st+3 (items 0 .. 5)
        if (last && (kt == 3 || kt == 6) && !(VMC_SPLIT_ABLATE & 2)) {
          // behind the three pieces of a row half this wave issued three ring pieces: those may stay in flight.
          // The staging area is this wave's own: no barrier, its vmcnt is the whole hand-over.
          const int hf = kt == 3 ? 0 : 1;
          asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
          gv[hf] = first_layer(ncoef[hf], s_mine[hf * 192], s_mine[hf * 192 + 64], s_mine[hf * 192 + 128]);
          asm volatile("" ::: "memory");           // (the reads stay in front of the next DMA into this area)
        }
        if (last && (kt == 0 || kt == 3) && !(VMC_SPLIT_ABLATE & 2)) {
          const int hf = kt == 0 ? 0 : 1;
          GATHER_DMA3(nsrc[hf].zb, nsrc[hf].wa, nsrc[hf].wb, __builtin_amdgcn_readfirstlane(stage_lds + hf * 3072), 64 * to);
        }
        if (kt < 6 && !(VMC_SPLIT_ABLATE & 4)) DMA_PIECE(kt);
        const u32x4 h = wh[kt & 1], m = wm[kt & 1], lo = wl[kt & 1];
        // smallest terms first
        if (!(VMC_SPLIT_ABLATE & 16)) {
        acc0 = mfma_bf16(lo, Xh[0][kt], acc0); acc1 = mfma_bf16(lo, Xh[1][kt], acc1);
        acc0 = mfma_bf16(h, Xl[0][kt], acc0); acc1 = mfma_bf16(h, Xl[1][kt], acc1);
        acc0 = mfma_bf16(m, Xm[0][kt], acc0); acc1 = mfma_bf16(m, Xm[1][kt], acc1);
        acc0 = mfma_bf16(m, Xh[0][kt], acc0); acc1 = mfma_bf16(m, Xh[1][kt], acc1);
        acc0 = mfma_bf16(h, Xm[0][kt], acc0); acc1 = mfma_bf16(h, Xm[1][kt], acc1);
        acc0 = mfma_bf16(h, Xh[0][kt], acc0); acc1 = mfma_bf16(h, Xh[1][kt], acc1);
        } else { acc0[0] += __uint_as_float(h[0] ^ Xh[0][kt][0]); acc1[0] += __uint_as_float(m[0] ^ lo[0] ^ Xm[1][kt][1] ^ Xl[0][kt][2]); }
        if (kt == 1 && to > 0) epilogue(to - 1, pacc0, pacc1, pwo);      // of the previous output tile, under these MFMAs
        // the operand split for the NEXT layer (or row tile), k-step by k-step inside the LAST stage of this layer:
        // item kt was the last reader of X[.][kt], and unit tiles 2 kt, 2 kt + 1 <= 13 of `out` are final (the
        // deferred epilogue of tile 14 runs above, tile 15 is this stage) -- 88 VALU instructions per item into the
        // MFMA shadows instead of a 704-instruction burst between two layers; k-step 7 follows behind the stage
        if (to == NT - 1 && kt < KT - 1 && !(VMC_SPLIT_ABLATE & 1)) {
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) split8(out[hf][2 * kt], out[hf][2 * kt + 1], Xh[hf][kt], Xm[hf][kt], Xl[hf][kt]);
        }
        if (kt == 1 || kt == 5) __builtin_amdgcn_sched_barrier(0);       // (keeps those two pieces of filler where they are)
      }
      if (!(VMC_SPLIT_ABLATE & 4)) advance();
      slot = nslot;
      bias = bias_next;
      if (last && !(VMC_SPLIT_ABLATE & 2)) { out[0][to] = gv[0]; out[1][to] = gv[1]; }
      else if (last) { out[0][to] = acc0; out[1][to] = acc1; }
      if (to + 1 < NT) { pacc0 = acc0; pacc1 = acc1; pwo = wo; }
      else {
#ifndef VMC_TAIL_NO_SETTLE
        vmc_mfma_settle(acc0, acc1);             // (read right behind the last MFMA, through inline asm: common.hpp)
#endif
        epilogue(to, acc0, acc1, wo);
      }
    }
  };

  for (;;) {
    const int next_tile = tile + gridDim.x;
    const bool has_next = next_tile * 128 < n_rows;               // workgroup-uniform
    const int nt_safe = has_next ? next_tile : tile;
    Desc nxt[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) { nxt[hf] = describe(nt_safe, hf); nsrc[hf] = source(nt_safe, hf); ncoef[hf] = nxt[hf].coef; }
    part[0] = part[1] = 0.f;
    for (int l = 0; l + 1 < n_hidden; ++l) {
      layer(bool_c<false>(), l);
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        constexpr int kt = KT - 1;               // (k-steps 0 .. 6 were split inside the layer's last stage)
        if (!(VMC_SPLIT_ABLATE & 1)) split8(out[hf][2 * kt], out[hf][2 * kt + 1], Xh[hf][kt], Xm[hf][kt], Xl[hf][kt]);
        else Xh[hf][kt][0] ^= __float_as_uint(out[hf][2 * kt][0] + out[hf][2 * kt + 1][3]);
      }
    }
    layer(bool_c<true>(), n_hidden - 1);      // out[] now holds the first-layer activations of the next row tile
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
    for (int hf = 0; hf < 2; ++hf) {
      constexpr int kt = KT - 1;
      if (!(VMC_SPLIT_ABLATE & 1)) split8(out[hf][2 * kt], out[hf][2 * kt + 1], Xh[hf][kt], Xm[hf][kt], Xl[hf][kt]);
      else Xh[hf][kt][0] ^= __float_as_uint(out[hf][2 * kt][0] + out[hf][2 * kt + 1][3]);
    }
  }
  // the prefetched fragments of the stage that is never multiplied, and my DMA still in flight
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(wh[0]), "+v"(wm[0]), "+v"(wl[0]) :: "memory");
#undef DMA_PIECE
}

static bool split_ring_on() { const char* e = getenv("CGS_VMC_SPLIT_RING"); return !(e && atoi(e) == 0); }

hipError_t launch_tail16_split(hipStream_t s, const TailArgs& a, const unsigned* p16s, bool ratio_mode) {
  if (a.n_rows <= 0) return hipSuccess;
  const int tiles = (a.n_rows + 127) / 128;
  const int persistent = a.num_cus > 0 ? a.num_cus : 256;
  const dim3 grid(tiles < persistent ? tiles : persistent), block(256);
  hipError_t e;
  // round 5: the weight stream through a per-CU LDS ring (k_tail16r); ring + bias / w_out image + gather staging
  // (networks whose bias image does not fit next to the ring -- more than 38 H x H layers -- keep k_tail16s)
  const size_t rlds = (size_t)SPLIT_RING * SPLIT_STAGE_BYTES + (size_t)(a.n_hidden + 1) * 1024 + 4 * 6144;
  if (split_ring_on() && a.n_hidden > 0 && rlds <= 160 * 1024) {
    if (ratio_mode) {
      e = hipFuncSetAttribute((const void*)k_tail16r<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL((k_tail16r<true>), grid, block, rlds, s, a, p16s);
    } else {
      e = hipFuncSetAttribute((const void*)k_tail16r<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL((k_tail16r<false>), grid, block, rlds, s, a, p16s);
    }
    return hipGetLastError();
  }
  const size_t lds = 0;
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
