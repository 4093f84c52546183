// hostcheck.cpp -- the host-side planners of libcgsvmc_hip.so (plan.hpp: parameter layouts, vmc_create's
// shape / LDS validation, convolution group / band / slice pickers, the sampler's LDS plan, split-K and
// XCD block-order maps, SR tile schedules, buffer sizes) driven over a grid of shapes on the CPU under
// AddressSanitizer + UndefinedBehaviourSanitizer (GPU sanitizers are not available on this pool):
//     make -C cgs_vmc_amd/csrc hostcheck        (g++ -fsanitize=address,undefined; tests/test_hostcheck.py)
// Every index a kernel derives from these plans is re-derived here into REAL arrays of the planned
// size, so that an offset one past a buffer is an ASan report and an overflowing product a UBSan one.
// The grid contains all five BASELINE configurations and the limits of include/cgsvmc.h (512 / 4096
// units, 64 filters, kernel 9, 32 x 32 lattices, 1023-site chains).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "plan.hpp"

static long long g_checks = 0, g_shapes = 0, g_rejected = 0, g_general = 0;
#define CHECK(cond)                                                                      \
  do {                                                                                   \
    ++g_checks;                                                                          \
    if (!(cond)) { fprintf(stderr, "hostcheck FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); abort(); } \
  } while (0)

static const int kCus[] = {256, 304, 64, 1};

static vmc_desc dense_desc(int ansatz, int n, int b, int L, int h, int act, int oact) {
  vmc_desc d;
  memset(&d, 0, sizeof(d));
  d.n_sites = n; d.batch_size = b; d.num_layers = L; d.layer_size = h;
  d.nonlinearity = act; d.output_activation = oact; d.ansatz = ansatz;
  return d;
}

// every piece of the parameter vector lands in [0, P), the pieces are disjoint and cover it
static void check_layout(const DescPlan& p, long long N, long long H) {
  const ParamLayout& lay = p.lay;
  const long long P = p.P;
  CHECK(P > 0);
  if (P > 6000000) {       // too large to mark: the pieces are consecutive, so the ends tell everything
    const long long last = p.rbm ? plan_off_b(lay, H, lay.n_hh) + H : lay.off_bout + 1;
    CHECK(last == P);
    CHECK(lay.off_w1 >= 0 && lay.off_b1 == lay.off_w1 + N * H && lay.off_h0 == lay.off_b1 + H);
    return;
  }
  std::vector<unsigned char> mark((size_t)P, 0);
  auto piece = [&](long long off, long long n) {
    CHECK(off >= 0 && off + n <= P);
    for (long long i = 0; i < n; ++i) { CHECK(mark[(size_t)(off + i)] == 0); mark[(size_t)(off + i)] = 1; }
  };
  piece(lay.off_w1, N * H); piece(lay.off_b1, H);
  for (int l = 1; l <= lay.n_hh; ++l) { piece(plan_off_w(lay, H, l), H * H); piece(plan_off_b(lay, H, l), H); }
  if (p.rbm) { CHECK(lay.off_wout == -1); piece(lay.off_won, N); piece(lay.off_bout, 1); }
  else { CHECK(lay.off_won == -1); piece(lay.off_wout, H); piece(lay.off_bout, 1); }
  for (long long i = 0; i < P; ++i) CHECK(mark[(size_t)i] == 1);
}

// XCD-aware block order: a bijection onto (slice, tile); every XCD gets the same number of work items
// (+- 1 range), contiguous in slice-major order
static void check_block_maps() {
  for (int tiles : {1, 4, 8, 16, 20, 33, 44})
    for (int slices : {1, 5, 7, 8, 9, 12, 32}) {
      const int grid = plan_wgrad_grid(tiles, slices), W = tiles * slices;
      CHECK(grid % 8 == 0 && grid >= W && grid < W + 8 * 8);
      std::vector<int> hit((size_t)W, 0), per_xcd(8, 0), last(8, -1);
      for (int b = 0; b < grid; ++b) {
        const WgradBlock m = plan_wgrad_block(b, tiles, slices);
        if (m.slice < 0) continue;
        CHECK(m.slice < slices && m.tile >= 0 && m.tile < tiles);
        const int w = m.slice * tiles + m.tile;
        hit[(size_t)w] += 1;
        per_xcd[(size_t)(b & 7)] += 1;
        CHECK(w > last[(size_t)(b & 7)]);               // one XCD walks a contiguous, ascending range
        last[(size_t)(b & 7)] = w;
      }
      for (int v : hit) CHECK(v == 1);
      const int G = (W + 7) / 8;
      for (int x = 0; x < 8; ++x) CHECK(per_xcd[(size_t)x] <= G && (per_xcd[(size_t)x] == G || (x + 1) * G > W));
    }
}

// the batched weight-gradient kernel: tiles, K slices and their chunks, the partial-tile workspace and
// the (tile -> problem, m-tile, n-tile) walk of k_wgrad
// (out_in_tiles = false: the output layer's sums come from the back-propagation kernel's partials and are
// folded by extra workgroups of the launch -- fully_connected on the fused kernels)
static void check_wgrad(long long N, long long H, int n_hh, bool rbm, long long B, bool out_in_tiles = true) {
  struct Prob { int k_in, n_out, tile0; };
  std::vector<Prob> probs;
  int tile0 = 0;
  auto add = [&](int k_in, int n_out) { probs.push_back({k_in, n_out, tile0}); tile0 += plan_wgrad_tiles(k_in, n_out); };
  if (out_in_tiles) add(rbm ? (int)N : (int)H, 1);
  for (int l = 0; l < n_hh; ++l) add((int)H, (int)H);
  add((int)N, (int)H);
  const long long tiles = tile0;
  CHECK(tiles == plan_wgrad_total_tiles((int)N, (int)H, n_hh, rbm, out_in_tiles) && tiles >= 1);
  const int other = 1 + (out_in_tiles ? 0 : plan_wgrad_fold_blocks((int)H));
  if (!out_in_tiles) {   // wgrad_fold_out: every output (two sums x (H weights + bias)) in exactly one (block, lane)
    const int nb = plan_wgrad_fold_blocks((int)H), n_out = 2 * ((int)H + 1);
    CHECK(nb >= 1 && nb * WG_FOLD_OUT >= n_out && (nb - 1) * WG_FOLD_OUT < n_out && 4 * WG_FOLD_OUT == 512);
    const int n_wg = (int)((B + 15) / 16), per = (n_wg + 3) >> 2;
    int covered = 0;
    for (int gq = 0; gq < 4; ++gq) { const int i0 = gq * per, i1 = std::min(n_wg, i0 + per); covered += i1 > i0 ? i1 - i0 : 0; }
    CHECK(covered == n_wg);
  }
  // every tile belongs to exactly one problem and lies inside its output; together they cover it
  for (const Prob& p : probs) {
    const int tn = (p.n_out + WG_TN - 1) / WG_TN, nt = plan_wgrad_tiles(p.k_in, p.n_out);
    for (int t = p.tile0; t < p.tile0 + nt; ++t) {
      int pi = 0;
      for (size_t i = 0; i < probs.size(); ++i)
        if (t >= probs[i].tile0) pi = (int)i;
      CHECK(&probs[(size_t)pi] == &p);
      const int lt = t - p.tile0, tm = lt / tn, tnn = lt % tn;
      CHECK(tm * WG_TM < p.k_in && tnn * WG_TN < p.n_out);
    }
    CHECK((long long)((p.k_in + WG_TM - 1) / WG_TM) * tn == nt);
  }
  for (int cus : kCus)
    for (int forced : {0, 1, 3, 12, 100}) {
      const int s = plan_wgrad_slices(tiles, B, cus, other, forced);
      CHECK(s >= 1 && s <= WG_MAX_SPLIT);
      if (!forced) CHECK(s == 1 || tiles * s + other <= cus);
      const int kc = plan_wgrad_kchunk((int)B, s);
      CHECK(kc % WG_TK == 0 && (long long)kc * s >= B);
      CHECK((long long)kc * (s - 1) < B);                 // no slice is empty
      // the last float the last slice of the last tile writes (second product's ones row)
      const long long part = 2 * ((long long)WG_TM * WG_TN + WG_TN);      // WG_PART of grad.hip
      const long long last = ((tiles - 1) * s + (s - 1)) * part + (part - 1);
      CHECK(tiles == 0 || last < plan_wgrad_ws_floats(tiles, WG_MAX_SPLIT));
    }
}

static void check_sweep(const vmc_desc& d, const DescPlan& p) {
  if (p.conv || (p.wide && !p.wide_fast)) return;
  const int NT = p.Hp / 16;
  CHECK(NT == 4 || NT == 8 || NT == 12 || NT == 16 || NT == 24 || NT == 32);   // the instantiated tile counts
  const int NW = (NT == 16 || NT >= 24) ? 8 : 4;
  const size_t need = plan_sweep_lds_required(d.n_sites, p.Hp, p.n_hh, p.rbm != 0);
  CHECK(need <= PLAN_LDS_PER_CU);
  for (int no_w1l = 0; no_w1l < 2; ++no_w1l)
    for (int plain = 0; plain < 2; ++plain)
      for (int tuned = 0; tuned < 2; ++tuned) {
        const SweepPlan sp = plan_sweep(d.n_sites, NT, NW, p.n_hh, p.rbm != 0, no_w1l != 0, plain != 0, tuned != 0);
        CHECK(sp.ok == 1);
        CHECK(sp.lds <= PLAN_LDS_PER_CU && sp.lds >= (sp.w1l ? (size_t)0 : need));
        CHECK(sp.fast == 0 || sp.fast == 2 || sp.fast == 4);
        CHECK(!(sp.w1l && (NT > 16 || no_w1l)));
        CHECK(!(sp.fast && (!plain || !tuned)));
        if (sp.fast == 2) CHECK((d.n_sites + 3) / 4 <= 32);      // two Philox draws cover 256 sites
        if (sp.fast == 4) CHECK((d.n_sites + 3) / 4 <= 64);
        if (sp.w1l) CHECK(sp.lds == plan_sweep_lds_bytes(d.n_sites, p.Hp, p.n_hh, true, p.rbm != 0));
      }
  if (p.wide_fast && p.n_hh >= 1) {
    CHECK(plan_tail_lds_supported(p.Hp, p.n_hh));
    CHECK(plan_tail_lds_bytes(p.Hp, p.n_hh) <= PLAN_LDS_PER_CU);
  }
  // the eight-chain sampler (sweep8.hip): its shapes, its LDS, and the tile rule
  for (int no_w1l = 0; no_w1l < 2; ++no_w1l) {
    const Sweep8Plan s8 = plan_sweep8(d.n_sites, p.Hp, p.n_hh, no_w1l != 0);
    if (s8.ok) {
      CHECK(p.Hp == 128 || p.Hp == 256);
      CHECK(p.n_hh >= 1 && d.n_sites <= p.Hp && d.n_sites <= 256);        // one Philox site block per lane of a chain's group
      CHECK(s8.lds <= PLAN_LDS_PER_CU && s8.lds == plan_sweep8_lds_bytes(d.n_sites, p.Hp, p.n_hh, s8.w1l != 0));
      CHECK(!(s8.w1l && no_w1l));
      // every array of the kernel's LDS carve-up starts on a 16-byte boundary
      const size_t nst = (size_t)((d.n_sites + 3) & ~3);
      CHECK((8 * nst) % 4 == 0 && (2 * 8 * ((size_t)p.Hp + 16)) % 4 == 0 && ((size_t)p.Hp + 16) % 4 == 0 && ((size_t)p.Hp + 4) % 4 == 0);
    } else {
      CHECK(s8.lds == 0 && s8.w1l == 0);
    }
    for (int forced : {0, 8, 16}) {
      const int tile = plan_sweep_tile(d.batch_size, 256, s8.ok != 0, forced);
      CHECK(tile == 8 || tile == 16);
      if (!s8.ok || forced == 16) CHECK(tile == 16);
      if (s8.ok && forced == 8) CHECK(tile == 8);
      if (s8.ok && forced == 0) CHECK((tile == 8) == ((d.batch_size + 15) / 16 <= 128));
      if (tile == 8 && forced == 0) CHECK((d.batch_size + 7) / 8 <= 256);   // never more eight-chain tiles than CUs by the rule itself
    }
  }
}

static void dense_grid() {
  const int sites[] = {2, 4, 10, 16, 36, 100, 144, 256, 400, 1024};
  const int units[] = {1, 7, 32, 64, 80, 128, 200, 256, 257, 300, 384, 400, 512, 513, 1024, 4096, 4097};
  const int layers[] = {0, 1, 2, 3, 6};
  const long long batches[] = {1, 64, 1000, 4096, 32768};
  char msg[256];
  for (int ansatz : {VMC_ANSATZ_FULLY_CONNECTED, VMC_ANSATZ_RBM})
    for (int n : sites)
      for (int h : units)
        for (int L : layers)
          for (int act : {VMC_ACT_RELU, VMC_ACT_COS, VMC_ACT_TANH}) {
            const vmc_desc d = dense_desc(ansatz, n, 4096, L, h, act, VMC_ACT_EXP);
            DescPlan p;
            for (int wf = 0; wf < 2; ++wf) {
              const int rc = plan_desc(&d, wf != 0, &p, msg, sizeof(msg));
              ++g_shapes;
              if (rc != VMC_OK) {
                ++g_rejected;
                CHECK(msg[0] != 0 && (rc == VMC_ERR_INVALID || rc == VMC_ERR_UNSUPPORTED));
                if (ansatz == VMC_ANSATZ_FULLY_CONNECTED && L == 0) CHECK(rc == VMC_ERR_INVALID);
                continue;
              }
              CHECK(h <= 4096 && !(L == 0 && ansatz == VMC_ANSATZ_FULLY_CONNECTED));
              CHECK(p.Hp >= h && p.Hp % 64 == 0 && p.Hp - h < 128);
              CHECK(p.wide == (h > 256));
              CHECK(!p.wide_fast || (wf && h > 256 && h <= 512 && (p.Hp == 384 || p.Hp == 512)));
              CHECK(p.n_hh == (ansatz == VMC_ANSATZ_RBM ? L : L - 1) && p.lay.n_hh == p.n_hh);
              CHECK(p.P == plan_num_params_dense(ansatz, n, h, L));
              check_layout(p, n, h);
              check_sweep(d, p);
              for (long long b : batches) {
                check_wgrad(n, h, p.n_hh, p.rbm != 0, b);
                if (!p.rbm && !(p.wide && !p.wide_fast)) check_wgrad(n, h, p.n_hh, false, b, false);
              }
            }
          }
  // the limits of include/cgsvmc.h, and what lies one step beyond them
  DescPlan p;
  vmc_desc d = dense_desc(VMC_ANSATZ_FULLY_CONNECTED, 100, 4096, 3, 4097, VMC_ACT_RELU, VMC_ACT_EXP);
  CHECK(plan_desc(&d, true, &p, msg, sizeof(msg)) == VMC_ERR_UNSUPPORTED);
  d = dense_desc(VMC_ANSATZ_FULLY_CONNECTED, 100, 4096, 3, 513, VMC_ACT_COS, VMC_ACT_EXP);
  CHECK(plan_desc(&d, true, &p, msg, sizeof(msg)) == VMC_OK && p.wide && !p.wide_fast);     // the general path has every activation
  d = dense_desc(VMC_ANSATZ_FULLY_CONNECTED, 100, 4096, 3, 512, VMC_ACT_COS, VMC_ACT_EXP);
  CHECK(plan_desc(&d, true, &p, msg, sizeof(msg)) == VMC_OK && p.wide_fast && p.Hp == 512);
  CHECK(plan_desc(&d, false, &p, msg, sizeof(msg)) == VMC_OK && !p.wide_fast);
  d = dense_desc(VMC_ANSATZ_RBM, 100, 4096, 2, 256, VMC_ACT_RELU, VMC_ACT_TANH);
  CHECK(plan_desc(&d, true, &p, msg, sizeof(msg)) == VMC_ERR_INVALID);
  d = dense_desc(VMC_ANSATZ_FULLY_CONNECTED, 100, 4096, 3, 256, 7, VMC_ACT_EXP);
  CHECK(plan_desc(&d, true, &p, msg, sizeof(msg)) == VMC_ERR_INVALID);
  d = dense_desc(7, 100, 4096, 3, 256, 0, VMC_ACT_EXP);
  CHECK(plan_desc(&d, true, &p, msg, sizeof(msg)) == VMC_ERR_UNSUPPORTED);
  // BASELINE configs 1 - 5 (config 4 = config 3 per rank)
  const int cfg[4][4] = {{16, 64, 2, 32}, {36, 1024, 3, 128}, {100, 4096, 3, 256}, {256, 1024, 6, 256}};
  for (auto& c : cfg) {
    d = dense_desc(VMC_ANSATZ_FULLY_CONNECTED, c[0], c[1], c[2], c[3], VMC_ACT_RELU, VMC_ACT_EXP);
    CHECK(plan_desc(&d, true, &p, msg, sizeof(msg)) == VMC_OK && !p.wide);
  }
  d = dense_desc(VMC_ANSATZ_FULLY_CONNECTED, 100, 4096, 3, 256, VMC_ACT_RELU, VMC_ACT_EXP);
  CHECK(plan_desc(&d, true, &p, msg, sizeof(msg)) == VMC_OK && p.P == 157697);
  const SweepPlan sp = plan_sweep(100, 16, 8, 2, false, false, true, true);
  CHECK(sp.ok && sp.w1l && sp.fast == 2);                     // config 3: W1 in LDS, two Philox draws per lane
  // BASELINE configs 2 and 5 (1,024 chains per GPU) take eight-chain tiles, config 3 (4,096) sixteen-chain tiles
  const Sweep8Plan c2 = plan_sweep8(36, 128, 2, false), c5 = plan_sweep8(256, 256, 5, false);
  CHECK(c2.ok && c2.w1l && c5.ok && !c5.w1l);                 // 16 x 16 sites: W1 (266 KB) stays in L2
  CHECK(plan_sweep_tile(1024, 256, true, 0) == 8 && plan_sweep_tile(4096, 256, true, 0) == 16 && plan_sweep_tile(2048, 256, true, 0) == 8);
  CHECK(!plan_sweep8(300, 256, 2, false).ok && !plan_sweep8(100, 256, 0, false).ok && !plan_sweep8(100, 192, 2, false).ok);
}

// index of (site, channel) in a feature map; wrap of a periodic coordinate
static long long fmap_index(const ConvGeom& g, int site, int c) { return (long long)(c / 4) * g.GS + 4 * site + c % 4; }
static int wrap(int v, int n) { v %= n; return v < 0 ? v + n : v; }

static void check_conv(const vmc_desc& d, const DescPlan& p) {
  const ConvGeom& g = p.cg;
  const long long B = d.batch_size;
  const int KK = g.K * g.KW, CW = 16 * g.NCB;
  CHECK(g.N == g.D1 * g.D2 && g.GS >= 4 * g.N && g.GS % 64 == 0 && g.CS == 4 * g.NCB * g.GS);
  CHECK(g.NCB >= 1 && g.NCB <= CONV_MAX_NCB && g.F <= CW && g.n_conv >= 1 && g.n_conv <= CONV_MAX_LAYERS);
  CHECK(g.lo + g.hi == g.K - 1 && g.lo2 + g.hi2 == g.KW - 1);
  CHECK(B * g.CS < (1LL << 31));
  {  // the feature-map layout is injective into [0, CS)
    std::vector<unsigned char> mark((size_t)g.CS, 0);
    for (int s = 0; s < g.N; ++s)
      for (int c = 0; c < CW; ++c) {
        const long long i = fmap_index(g, s, c);
        CHECK(i >= 0 && i < g.CS && mark[(size_t)i] == 0);
        mark[(size_t)i] = 1;
      }
  }
  CHECK(p.P == plan_num_params_conv(g.n_conv, g.F, KK));
  {  // k_conv_pack: every theta element it reads is inside the parameter vector, every image element
     // it writes inside the planned image
    const long long p0 = (long long)KK * g.F + g.F, pl = (long long)KK * g.F * g.F + g.F, Q0 = (KK + 3) / 4;
    CHECK(plan_conv_w0_floats(g) == (long long)g.NCB * Q0 * 64);
    long long max_theta = -1;
    for (long long i = 0; i < (long long)g.NCB * Q0 * 64; ++i) {
      const int cb = (int)(i / (Q0 * 64)), q = (int)((i / 64) % Q0), lane = (int)(i % 64), m = lane & 15, gq = lane >> 4;
      const int tap = 4 * q + gq, co = 16 * cb + m;
      if (tap < KK && co < g.F) max_theta = std::max(max_theta, (long long)tap * g.F + co);
    }
    CHECK(max_theta < p0);
    for (int l = 0; l < g.n_conv; ++l) {
      const long long base = l == 0 ? (long long)KK * g.F : p0 + (long long)(l - 1) * pl + (long long)KK * g.F * g.F;
      CHECK(base + g.F <= p.P);
    }
    CHECK(plan_conv_bias_floats(g) == (long long)g.n_conv * 16 * g.NCB);
    const long long per_layer = (long long)KK * 256 * g.NCB * g.NCB;
    CHECK((long long)(g.n_conv - 1) * per_layer <= plan_conv_wf_floats(g));
    if (g.n_conv > 1) {
      const long long wlast = p0 + (long long)(g.n_conv - 2) * pl;     // [tap][cin][cout] of the last convolution
      CHECK(wlast + ((long long)(KK - 1) * g.F + (g.F - 1)) * g.F + (g.F - 1) < p.P);
    }
  }
  const int nw = plan_conv_waves(g);
  CHECK(nw == 4 || nw == 8);
  for (int one = 0; one < 2; ++one) {
    const size_t cap = plan_conv_lds_cap(g, one != 0);
    CHECK(cap == CONV_LDS_PER_WG || cap == PLAN_LDS_PER_CU);
    const int G = plan_conv_pick_group(g, nw, one != 0);
    CHECK(G >= 1 && G <= 64 && plan_conv_rows_lds(g, G) <= cap);
    for (int cus : kCus) {
      const int Gs = plan_conv_pick_sweep_group(g, B, cus, nw, one != 0);
      CHECK(Gs >= 1 && Gs <= 64 && Gs <= B && plan_conv_rows_lds(g, Gs) <= cap);
      const int grid = plan_conv_grid(g, B * 7, G, cus);
      CHECK(grid >= 1 && grid <= 2 * cus && (long long)grid * G <= B * 7 + G);
      const int sl = plan_conv_dw_slices(g, B, cus);
      CHECK(sl >= 1 && sl <= 256 && sl <= B);
      {  // grid z of k_conv_dw: the output block groups x item parts cover every block and every item
         // once, with at most PLAN_DW_MAX_ACC accumulator pairs per wave
        const int nco = plan_conv_dw_nco(g), parts = plan_conv_dw_parts(g.K, g.KW, g.NCB);
        CHECK(nco >= 1 && g.NCB % nco == 0 && parts >= 1 && plan_conv_dw_grid_z(g) == (g.NCB / nco) * parts);
        const int ni = KK + 1, tpw = (ni + PLAN_DW_WAVES * parts - 1) / (PLAN_DW_WAVES * parts);
        CHECK(tpw * g.NCB * nco <= PLAN_DW_MAX_ACC);
        std::vector<int> seen((size_t)ni, 0);
        for (int part = 0; part < parts; ++part)
          for (int i = 0; i < tpw; ++i)
            for (int w = 0; w < PLAN_DW_WAVES; ++w) {
              const int it = w + (part * tpw + i) * PLAN_DW_WAVES;
              if (it < ni) ++seen[(size_t)it];
            }
        for (int it = 0; it < ni; ++it) CHECK(seen[(size_t)it] == 1);
        CHECK((KK + 15) / 16 <= PLAN_DW_WAVES);       // the first layer's tap tiles: one per wave, one part
      }
      // weight-gradient workspace: the last element slice sl-1 / layer n_conv-1 / second sum writes
      const long long rows = (long long)KK * CW + 1, ws = plan_conv_dw_ws_floats(g, sl);
      const long long last = ((((long long)(sl - 1) * g.n_conv + (g.n_conv - 1)) * 2) + 1) * rows * CW + (rows * CW - 1);
      CHECK(last == ws - 1);
      // k_conv_dw_reduce's source index of the last weight and the last bias
      CHECK(((long long)(KK - 1) * CW + (g.F - 1)) * CW + (g.F - 1) < rows * CW);
      CHECK((long long)KK * CW * CW + (g.F - 1) < rows * CW);
    }
  }
  // weight-gradient bands: a band always exists, fits, and its padded numbering stays inside its maps
  for (int forced : {0, 1, 3}) {
    const int rb = plan_conv_dw_band(g, forced);
    CHECK(rb >= 1 && rb <= g.D1 && plan_conv_dw_lds(g, rb) <= PLAN_LDS_PER_CU);
    if (forced >= 1 && forced < g.D1) CHECK(rb <= forced);
    const int D2p = g.D2 + g.KW - 1, NPAD = (g.D1 + g.K - 1) * D2p;
    const int NQ = (rb * D2p + 3) & ~3, NIN = NQ + (g.K - 1) * D2p + g.KW;
    std::vector<int> s_map((size_t)NPAD), s_pos((size_t)g.N);
    for (int i = 0; i < NPAD; ++i) {
      const int p1 = i / D2p, p2 = i - p1 * D2p;
      s_map[(size_t)i] = wrap(p1 - g.lo, g.D1) * g.D2 + wrap(p2 - g.lo2, g.D2);
      CHECK(s_map[(size_t)i] >= 0 && s_map[(size_t)i] < g.N);
    }
    for (int i = 0; i < g.N; ++i) {
      const int a1 = i / g.D2;
      s_pos[(size_t)i] = a1 * D2p + (i - a1 * g.D2);
      CHECK(s_pos[(size_t)i] < NPAD);
    }
    // a band of rb rows starting at row r0: position q of the band + tap offset < NIN
    const int q_last = (rb - 1) * D2p + (g.D2 - 1);
    CHECK(q_last < NQ && q_last + (g.K - 1) * D2p + (g.KW - 1) < NIN);
    const int CWD = 16 * plan_conv_dw_nco(g);
    const size_t floats = ((size_t)NQ * CWD + (size_t)NIN * CW + NPAD + g.N + CW + 8 * CW);
    // the walk reads delta one quad past the end (dropped): NQ + 4 positions of CWD floats stay inside the LDS
    CHECK(((size_t)NQ + 4) * CWD <= floats);
    CHECK(floats * sizeof(float) == plan_conv_dw_lds(g, rb));
  }
}

static void conv_grid() {
  struct Lat { int x, y; };
  const Lat lats[] = {{4, 4}, {3, 5}, {6, 6}, {10, 10}, {16, 16}, {24, 24}, {30, 30}, {32, 32}, {33, 33}, {2, 50}};
  const int chains[] = {8, 40, 100, 1000, 1023, 1024};
  const int filters[] = {1, 4, 8, 16, 17, 32, 33, 48, 49, 64, 65};
  const long long batches[] = {1, 7, 1024, 4096, 100000};
  char msg[256];
  for (int ansatz : {VMC_ANSATZ_CONV_2D, VMC_ANSATZ_RES_NET_2D, VMC_ANSATZ_CONV_1D, VMC_ANSATZ_RES_NET_1D}) {
    const bool one_d = ansatz == VMC_ANSATZ_CONV_1D || ansatz == VMC_ANSATZ_RES_NET_1D;
    const bool resnet = ansatz == VMC_ANSATZ_RES_NET_2D || ansatz == VMC_ANSATZ_RES_NET_1D;
    const int n_lat = one_d ? (int)(sizeof(chains) / sizeof(chains[0])) : (int)(sizeof(lats) / sizeof(lats[0]));
    for (int li = 0; li < n_lat; ++li)
      for (int F : filters)
        for (int K = 1; K <= 10; ++K)
          for (int L : {0, 1, 2, 5, 16})
            for (long long B : batches) {
              vmc_desc d;
              memset(&d, 0, sizeof(d));
              d.ansatz = ansatz; d.batch_size = (int)B; d.num_layers = L; d.layer_size = F; d.kernel_size = K;
              d.nonlinearity = VMC_ACT_RELU; d.output_activation = VMC_ACT_EXP;
              if (one_d) { d.n_sites = chains[li]; }
              else { d.size_x = lats[li].x; d.size_y = lats[li].y; d.n_sites = d.size_x * d.size_y; }
              DescPlan p;
              const int rc = plan_desc(&d, true, &p, msg, sizeof(msg));
              ++g_shapes;
              if (rc != VMC_OK) {
                ++g_rejected;
                CHECK(msg[0] != 0);
                continue;
              }
              CHECK((resnet || L >= 1) && p.conv);
              if (p.conv_general) {       // beyond the fused kernels for one of their four reasons -- or sent here because the patch kernels beat them
                const bool refused = F > 64 || K > 9 || plan_conv_rows_lds(p.cg, 1) > PLAN_LDS_PER_CU || B * p.cg.CS >= (1LL << 31);
                CHECK(refused || plan_cgen_patch_routes(p.cg, B));
                if (!refused) {           // ... by preference only: CGS_VMC_CONV_GENERAL=0 keeps the fused kernels
                  DescPlan pf;
                  CHECK(plan_desc(&d, true, &pf, msg, sizeof(msg), -1) == VMC_OK && !pf.conv_general);
                }
                if (plan_cgen_patch_ok(p.cg, B)) {      // the patch kernels' LDS plan: within a CU's, every piece inside it
                  const size_t lds = plan_cgen_patch_lds_bytes(p.cg);
                  const int nc = p.cg.n_conv;
                  CHECK(lds <= PLAN_CGEN_PATCH_LDS && F <= 16 && nc >= 2 && nc <= 9);
                  CHECK(plan_cgen_patch_side(p.cg, nc - 1, 0) <= p.cg.D1 && plan_cgen_patch_side(p.cg, nc - 1, 1) <= p.cg.D2);
                  CHECK(2LL * plan_cgen_patch_side(p.cg, nc - 1, 0) * plan_cgen_patch_side(p.cg, nc - 1, 1) <= 32768);   // 16-bit marks: the last index is 2 s1 s2 - 1
                  CHECK(!resnet || !plan_cgen_patch_routes(p.cg, B));      // residual networks come here only when the fused kernels refuse them
                }
                CHECK((long long)p.cg.N * plan_cgen_lda(p.cg) < (1LL << 28));
                // the band kernel of the general path (<= 16 filters, 2 .. 7 taps): a band with its halo fits its LDS
                // budget, the bands cover the lattice, the staged index stays in 32 bits
                if (plan_cgen_band_ok(p.cg)) {
                  const int bh = plan_cgen_band_rows(p.cg);
                  CHECK(F <= 64 && K >= 2 && K <= 7 && bh >= 1 && bh <= p.cg.D1);
                  CHECK(p.cg.K * p.cg.KW * 4 * plan_cgen_band_ncb(p.cg) <= PLAN_CGEN_BAND_MAX_FRAGS && plan_cgen_band_ncb(p.cg) <= 4);
                  CHECK(plan_cgen_band_lds_bytes(p.cg, false) <= (size_t)(plan_cgen_band_ncb(p.cg) > 1 ? PLAN_CGEN_BAND_LDS_WIDE : PLAN_CGEN_BAND_LDS) &&
                        plan_cgen_band_lds_bytes(p.cg, true) <= plan_cgen_band_lds_bytes(p.cg, false));
                  const int nb = (p.cg.D1 + bh - 1) / bh;
                  CHECK((long long)nb * bh >= p.cg.D1 && (long long)(nb - 1) * bh < p.cg.D1);
                  CHECK((long long)(bh + p.cg.K - 1) * (p.cg.D2 + p.cg.KW - 1) * 16 * plan_cgen_band_ncb(p.cg) < (1LL << 24));
                  for (long long rows : {1LL, 7LL, 32LL, 1000LL, 100000LL})       // thinner bands for few rows, never beyond the LDS band
                    for (long long cap : {256LL, 512LL, 1024LL}) {
                      const int b2 = plan_cgen_band_rows_for(p.cg, rows, cap);
                      CHECK(b2 >= 1 && b2 <= bh);
                      CHECK(plan_cgen_band_lds_bytes(p.cg, false, b2) <= plan_cgen_band_lds_bytes(p.cg, false));
                      if (rows * ((p.cg.D1 + bh - 1) / bh) >= cap) CHECK(b2 == bh);
                    }
                } else {
                  CHECK(F > 64 || K > 7 || K < 2 || p.cg.K * p.cg.KW * 4 * plan_cgen_band_ncb(p.cg) > PLAN_CGEN_BAND_MAX_FRAGS ||
                        plan_cgen_band_rows(p.cg) < 1);
                }
                ++g_general;
                continue;
              }
              CHECK(F <= 64 && K <= 9);
              CHECK(plan_conv_rows_lds(p.cg, 1) <= PLAN_LDS_PER_CU);
              check_conv(d, p);
            }
  }
  // named limits: 64 filters and kernel 9 are in, 65 and 10 are out; the bench workloads are in
  vmc_desc d;
  DescPlan p;
  memset(&d, 0, sizeof(d));
  d.ansatz = VMC_ANSATZ_CONV_2D; d.batch_size = 4096; d.num_layers = 5; d.layer_size = 32; d.kernel_size = 7;
  d.size_x = d.size_y = 10; d.n_sites = 100; d.output_activation = VMC_ACT_EXP;
  CHECK(plan_desc(&d, true, &p, msg, sizeof(msg)) == VMC_OK && p.cg.NCB == 2);
  d.layer_size = 64; d.kernel_size = 9;
  CHECK(plan_desc(&d, true, &p, msg, sizeof(msg)) == VMC_OK && p.cg.NCB == 4 && plan_conv_tab(p.cg) == 16);
  CHECK(plan_conv_dw_nco(p.cg) == 1 && plan_conv_dw_parts(9, 9, 4) == 2 && plan_conv_dw_grid_z(p.cg) == 8);
  CHECK(plan_conv_dw_nco(5, 5, 2) == 2 && plan_conv_dw_parts(5, 5, 2) == 1);      // the measured 32-filter kernels are unchanged
  CHECK(plan_conv_dw_nco(5, 5, 1) == 1 && plan_conv_dw_parts(5, 5, 1) == 1 && plan_conv_dw_parts(9, 9, 1) == 1);
  CHECK(!p.conv_general);
  d.layer_size = 65;                                                           // beyond: the general path (conv_general.hip)
  CHECK(plan_desc(&d, true, &p, msg, sizeof(msg)) == VMC_OK && p.conv_general);
  d.layer_size = 1025;
  CHECK(plan_desc(&d, true, &p, msg, sizeof(msg)) == VMC_ERR_UNSUPPORTED);
  d.layer_size = 16; d.kernel_size = 10;
  CHECK(plan_desc(&d, true, &p, msg, sizeof(msg)) == VMC_OK && p.conv_general && p.P == plan_num_params_conv(5, 16, 100));
  d.kernel_size = 32;
  CHECK(plan_desc(&d, true, &p, msg, sizeof(msg)) == VMC_ERR_UNSUPPORTED);
  d.kernel_size = 5;
  CHECK(plan_desc(&d, true, &p, msg, sizeof(msg), true) == VMC_OK && p.conv_general);      // forced (CGS_VMC_CONV_GENERAL=1)
  CHECK(plan_cgen_band_ok(p.cg) && plan_cgen_band_rows(p.cg) == 10);           // 16 filters 5 x 5 on 10 x 10: the band kernel, one band
  {  // 36 x 36 x 16 filters, 5 x 5 (bench workload heisenberg36x36_conv3x16k5_b32): three bands of 12 lattice rows
    vmc_desc b = d; DescPlan q;
    b.layer_size = 16; b.num_layers = 3; b.size_x = b.size_y = 36; b.n_sites = 1296; b.batch_size = 32;
    CHECK(plan_desc(&b, true, &q, msg, sizeof(msg)) == VMC_OK && q.conv_general && plan_cgen_band_ok(q.cg));
    CHECK(plan_cgen_band_rows(q.cg) == 12 && plan_cgen_band_lds_bytes(q.cg, false) == 16u * 40u * 64u);
    // the patch sampler: boxes of 5, 9, 13 sites per axis; spins 1,296 + fragments 2 x 25 x 256 + biases 48 + boxes
    // 2 x 16 (25 + 81 + 169) + two windows of 17 x 17 x 16 floats + 1,296 16-bit marks + 1,296 uniforms
    CHECK(plan_cgen_patch_ok(q.cg, 32) && plan_cgen_patch_pays(q.cg) && plan_cgen_patch_side(q.cg, 2, 0) == 13);
    CHECK(plan_cgen_patch_lds_bytes(q.cg) == 4u * (1296u + 12800u + 48u + 8800u + 2u * 4624u + 648u + 1296u) + 256u);
    {
      DescPlan t; vmc_desc s = b;
      s.size_x = s.size_y = 12; s.n_sites = 144;                       // the last box (13) would meet itself around the torus
      CHECK(plan_desc(&s, true, &t, msg, sizeof(msg), true) == VMC_OK && t.conv_general && !plan_cgen_patch_ok(t.cg, 32));
      s.size_x = s.size_y = 13; s.n_sites = 169;                       // fits exactly -- and covers the lattice: does not pay
      CHECK(plan_desc(&s, true, &t, msg, sizeof(msg), true) == VMC_OK && plan_cgen_patch_ok(t.cg, 32) && !plan_cgen_patch_pays(t.cg));
      s = b; s.num_layers = 1;                                         // one convolution: nothing to keep
      CHECK(plan_desc(&s, true, &t, msg, sizeof(msg)) == VMC_OK && !plan_cgen_patch_ok(t.cg, 32));
      // routing: a shape the fused kernels take goes to the general path where the boxes are at most a fifth of a forward
      s = b; s.size_x = s.size_y = 24; s.n_sites = 576; s.num_layers = 2; s.batch_size = 256;     // 2 (25 + 81) of 2 x 576: 18 %
      CHECK(plan_desc(&s, true, &t, msg, sizeof(msg)) == VMC_OK && t.conv_general && plan_cgen_patch_routes(t.cg, 256));
      CHECK(plan_desc(&s, true, &t, msg, sizeof(msg), -1) == VMC_OK && !t.conv_general);         // CGS_VMC_CONV_GENERAL=0
      s.size_x = s.size_y = 16; s.n_sites = 256;                                                  // 2 (25 + 81) of 2 x 256: 41 %
      CHECK(plan_desc(&s, true, &t, msg, sizeof(msg)) == VMC_OK && !t.conv_general && plan_cgen_patch_ok(t.cg, 256) && !plan_cgen_patch_routes(t.cg, 256));
      s = b; s.layer_size = 32; s.kernel_size = 3;                     // two channel blocks
      CHECK(plan_desc(&s, true, &t, msg, sizeof(msg)) == VMC_OK && t.conv_general && !plan_cgen_patch_ok(t.cg, 32));
    }
    // the sampler's chain groups: one where the launches are latency, two for the one-workgroup-per-CU band kernel
    CHECK(plan_cgen_sweep_groups(q.cg, 32, 256) == 1);
    b.layer_size = 64; b.kernel_size = 3;
    CHECK(plan_desc(&b, true, &q, msg, sizeof(msg)) == VMC_OK && q.conv_general && plan_cgen_band_ok(q.cg));
    CHECK(plan_cgen_sweep_groups(q.cg, 32, 256) == 2 && plan_cgen_sweep_groups(q.cg, 1, 256) == 1);
    // ... and for the GEMM form where the last round of row tiles is mostly empty (800 tiles: 3.125 rounds of 256 CUs)
    b.layer_size = 128; b.size_x = b.size_y = 10; b.n_sites = 100; b.batch_size = 1024;
    CHECK(plan_desc(&b, true, &q, msg, sizeof(msg)) == VMC_OK && q.conv_general && !plan_cgen_band_ok(q.cg));
    CHECK(plan_cgen_sweep_groups(q.cg, 1024, 256) == 2);       // 800 tiles
    CHECK(plan_cgen_sweep_groups(q.cg, 4096, 256) == 1);       // 3,200 tiles: 12.5 rounds, 0.96 full
    CHECK(plan_cgen_sweep_groups(q.cg, 256, 256) == 1);        // 200 tiles: less than a round
    CHECK(plan_cgen_sweep_groups(q.cg, 1280, 256) == 1);       // 1,000 tiles: 3.9 rounds
  }
  d.kernel_size = 10;
  d.kernel_size = 5; d.size_x = 9;
  CHECK(plan_desc(&d, true, &p, msg, sizeof(msg)) == VMC_ERR_INVALID);        // size_x * size_y != num_sites
  d.size_x = 10;
  CHECK(plan_desc(&d, true, &p, msg, sizeof(msg)) == VMC_OK && p.P == plan_num_params_conv(5, 16, 25));
  CHECK(plan_conv_pick_group(p.cg, 4) == 5);                                 // G = 5 on the 10 x 10 lattice
}

static void sr_schedules() {
  unsigned state = 12345u;
  auto rnd = [&](int lo, int hi) { state = state * 1664525u + 1013904223u; return lo + (int)((state >> 8) % (unsigned)(hi - lo + 1)); };
  for (int trial = 0; trial < 2000; ++trial) {
    const int n = rnd(1, RD_MAXB);
    int K[RD_MAXB], M[RD_MAXB], order[RD_MAXB], first[RD_MAXB + 1];
    for (int j = 0; j < n; ++j) { K[j] = rnd(1, 512); M[j] = rnd(1, 300000); }
    const int tiles = plan_rowdot_schedule(K, M, n, order, first);
    std::vector<int> seen((size_t)n, 0);
    long long sum = 0;
    for (int j = 0; j < n; ++j) {
      CHECK(order[j] >= 0 && order[j] < n);
      seen[(size_t)order[j]] += 1;
      if (j > 0) {
        CHECK(K[order[j]] <= K[order[j - 1]]);
        if (K[order[j]] == K[order[j - 1]]) CHECK(order[j] > order[j - 1]);   // stable
      }
      CHECK(first[j] == sum);
      sum += (M[order[j]] + RD_TM - 1) / RD_TM;
    }
    for (int v : seen) CHECK(v == 1);
    CHECK(first[n] == sum && tiles == sum);
  }
  for (int R : {1, 63, 64, 65, 4096, 204800})
    for (int cus : kCus) {
      const int s = plan_sr_wsum_slices(R, cus);
      CHECK(s >= 1 && s <= cus && (s == 1 || (long long)(s - 1) * 64 < R));
    }
}

int main() {
  check_block_maps();
  dense_grid();
  conv_grid();
  sr_schedules();
  printf("hostcheck ok: %lld shapes (%lld rejected by plan_desc, %lld convolutional ones on the general path), %lld assertions\n", g_shapes, g_rejected, g_general, g_checks);
  return 0;
}
