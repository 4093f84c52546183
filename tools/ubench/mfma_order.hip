// Probe (round 6, before k_sweep8): two facts about the gfx950 matrix cores that the 8-chain sampler rests on.
//  (1) v_mfma_f32_16x16x4_f32 accumulates its four k slots as a sequential chain of fused multiply-adds in k order
//      (c -> k0 -> k1 -> k2 -> k3): then four v_mfma_f32_4x4x1_16B_f32 (k = 1 each) in that order give the same bits,
//      and a sampler on the 4x4x1 shape can be bit-identical to k_sweep16.
//  (2) the operand broadcast controls of v_mfma_f32_4x4x1_16B_f32: CBSZ = 3 / ABID = j (block j of each group of eight
//      blocks supplies the A operand of all eight) and BLGP = 1 / 2 (lanes 0-31 / 32-63 of the B operand serve both halves).
// Prints PASS / FAIL per fact and the first mismatch.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_order.hip -o /tmp/mfma_order && /tmp/mfma_order
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k16(const float* a, const float* b, const float* c, float* d) {
  const int l = threadIdx.x;
  f32x4 acc;
  for (int v = 0; v < 4; ++v) acc[v] = c[l * 4 + v];
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], acc, 0, 0, 0);
  for (int v = 0; v < 4; ++v) d[l * 4 + v] = acc[v];
}

// one 4x4x1 with the given controls (template immediates)
template <int CBSZ, int ABID, int BLGP>
__global__ void k4(const float* a, const float* b, const float* c, float* d) {
  const int l = threadIdx.x;
  f32x4 acc;
  for (int v = 0; v < 4; ++v) acc[v] = c[l * 4 + v];
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, CBSZ, ABID, BLGP);
  for (int v = 0; v < 4; ++v) d[l * 4 + v] = acc[v];
}

static float* dev(const std::vector<float>& h) {
  float* p; (void)hipMalloc(&p, h.size() * 4); (void)hipMemcpy(p, h.data(), h.size() * 4, hipMemcpyHostToDevice); return p;
}
static unsigned bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }

template <int CBSZ, int ABID, int BLGP>
static bool check4(const std::vector<float>& a, const std::vector<float>& b, const std::vector<float>& c, const char* what) {
  std::vector<float> d(256);
  float *da = dev(a), *db = dev(b), *dc = dev(c), *dd = dev(d);
  hipLaunchKernelGGL((k4<CBSZ, ABID, BLGP>), dim3(1), dim3(64), 0, 0, da, db, dc, dd);
  (void)hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int blk = l / 4, n = l % 4;
    // model: A of block blk comes from block (blk / 2^CBSZ) * 2^CBSZ + ABID; B lane from the half BLGP names
    const int ablk = CBSZ ? (blk >> CBSZ << CBSZ) + ABID : blk;
    int bl = l;
    if (BLGP == 1) bl = l % 32;
    if (BLGP == 2) bl = 32 + l % 32;
    for (int v = 0; v < 4; ++v) {
      const float want = fmaf(a[ablk * 4 + v], b[bl], c[l * 4 + v]);
      if (bits(want) != bits(d[l * 4 + v])) {
        if (!bad) printf("  %s: first mismatch lane %d (block %d col %d) row %d: got %.9g want %.9g\n", what, l, blk, n, v, d[l * 4 + v], want);
        ++bad;
      }
    }
  }
  printf("%s: %s (%d of 256 differ)\n", what, bad ? "FAIL" : "PASS", bad);
  return !bad;
}

int main() {
  std::vector<float> a(64), b(64), c(256), d(256);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f * 2.f - 1.f; };
  int total_bad[4] = {0, 0, 0, 0};
  for (int trial = 0; trial < 50; ++trial) {
    for (auto& x : a) x = rnd() * (1.f + 1000.f * (trial % 3));
    for (auto& x : b) x = rnd();
    for (auto& x : c) x = rnd() * (trial % 2 ? 1e-3f : 10.f);
    float *da = dev(a), *db = dev(b), *dc = dev(c), *dd = dev(d);
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
    (void)hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l)
      for (int v = 0; v < 4; ++v) {
        const int j = l % 16, i = 4 * (l / 16) + v;    // D: column j, row i;  A[i][k] = a[k * 16 + i], B[k][j] = b[k * 16 + j]
        float p[4];
        for (int k = 0; k < 4; ++k) p[k] = 0.f;
        float seq = c[l * 4 + v], rev = c[l * 4 + v];
        for (int k = 0; k < 4; ++k) seq = fmaf(a[k * 16 + i], b[k * 16 + j], seq);
        for (int k = 3; k >= 0; --k) rev = fmaf(a[k * 16 + i], b[k * 16 + j], rev);
        // pairs: (k0, k1) chain and (k2, k3) chain summed;  exact sum rounded once (double is exact enough for a flag)
        const float pa = fmaf(a[16 + i], b[16 + j], a[i] * b[j]), pb = fmaf(a[48 + i], b[48 + j], a[32 + i] * b[32 + j]);
        const float pair = c[l * 4 + v] + (pa + pb);
        double ex = c[l * 4 + v];
        for (int k = 0; k < 4; ++k) ex += (double)a[k * 16 + i] * (double)b[k * 16 + j];
        const float once = (float)ex;
        const float got = d[l * 4 + v];
        total_bad[0] += bits(got) != bits(seq);
        total_bad[1] += bits(got) != bits(rev);
        total_bad[2] += bits(got) != bits(pair);
        total_bad[3] += bits(got) != bits(once);
      }
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dc); (void)hipFree(dd);
  }
  printf("16x16x4 f32 over 50 x 256 outputs: differs from  k-ascending fma chain %d,  k-descending chain %d,  pairwise %d,  single rounding %d\n",
         total_bad[0], total_bad[1], total_bad[2], total_bad[3]);
  printf("fact 1 (16x16x4 == ascending fma chain): %s\n", total_bad[0] == 0 ? "PASS" : "FAIL");
  for (auto& x : a) x = rnd();
  for (auto& x : b) x = rnd();
  for (auto& x : c) x = rnd();
  bool ok = true;
  ok &= check4<0, 0, 0>(a, b, c, "4x4x1 plain");
  ok &= check4<3, 0, 0>(a, b, c, "4x4x1 cbsz=3 abid=0");
  ok &= check4<3, 5, 0>(a, b, c, "4x4x1 cbsz=3 abid=5");
  ok &= check4<0, 0, 1>(a, b, c, "4x4x1 blgp=1");
  ok &= check4<0, 0, 2>(a, b, c, "4x4x1 blgp=2");
  ok &= check4<3, 6, 2>(a, b, c, "4x4x1 cbsz=3 abid=6 blgp=2");
  printf("fact 2 (CBSZ / ABID / BLGP model): %s\n", ok ? "PASS" : "FAIL");
  return 0;
}
