// Host side and instantiations of the narrow kernels (mimo_narrow_kernel.h): slot loops of Dz <= 4 and the table-driven loops;
// the grouped loops are instantiated in mimo_narrow_grouped.hip.
#include "mimo_narrow_kernel.h"

namespace mimo {

// ------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------
// component slots per lane the kernels are instantiated for (4 V >= K); every multiple of 4 is there, so 4 V never exceeds
// the 16-padded component count of the partial block
static const int kNarrowV[] = {1, 2, 3, 4, 6, 8, 10, 12, 13, 14, 16, 18, 20, 22, 24, 25, 26, 28, 30, 32,
                               36, 40, 44, 48, 52, 56, 60, 64};      // (above 32: mimo_narrow_big.hip, at most two contraction steps)
int narrow_v(int K) {
  const int need = (K + 3) / 4;
  for (int v : kNarrowV) if (v >= need) return v;
  return 0;
}
int narrow_nsf(int F) { return (F + 3) / 4; }

size_t narrow_lds_bytes(int V, int NSF, int ZS) {
  return sizeof(double) * ((size_t)(NSF * V + kNarrowPF) * 16 + kExpTab + (size_t)4 * 16 * ZS + 4) + sizeof(uint32_t) * 4 * (size_t)NSF;
}


// instantiated: up to 16 features (NSF <= 4) for Dz <= 4 (ZI = 1: the 16 x Dz <= 64 elements of a step in one load per lane) with
// every slot count; more features (the table-driven loops) for Dz = 5 .. 16 (ZI = 4) while the V NSF accumulators fit: V NSF <= 96
template <int V>
static narrow_fn pick_narrow_nsf(int nsf, int gibbs, int zi) {
  switch (nsf) {
#define MIMO_NN(n) case n: return pick_narrow_mode<V, n>(gibbs, zi);
    MIMO_NN(1) MIMO_NN(2) MIMO_NN(3) MIMO_NN(4)       // (five steps and more: mimo_narrow_table.hip)
#undef MIMO_NN
  }
  return nullptr;
}
narrow_fn pick_narrow_table(int V, int nsf, int gibbs, int zi);     // mimo_narrow_table.hip
narrow_fn pick_narrow_big(int V, int nsf, int gibbs, int zi);       // mimo_narrow_big.hip
static narrow_fn pick_narrow(int V, int nsf, int gibbs, int zi) {
  if (nsf > 4) return V > 32 ? nullptr : pick_narrow_table(V, nsf, gibbs, zi);
  if (V > 32) return (nsf <= 2 || gibbs == 1) ? pick_narrow_big(V, nsf, gibbs, zi) : nullptr;
  switch (V) {
#define MIMO_NV(v) case v: return pick_narrow_nsf<v>(nsf, gibbs, zi);
    MIMO_NV(1) MIMO_NV(2) MIMO_NV(3) MIMO_NV(4) MIMO_NV(6) MIMO_NV(8) MIMO_NV(10) MIMO_NV(12) MIMO_NV(13) MIMO_NV(14)
    MIMO_NV(16) MIMO_NV(18) MIMO_NV(20) MIMO_NV(22) MIMO_NV(24) MIMO_NV(25) MIMO_NV(26) MIMO_NV(28) MIMO_NV(30) MIMO_NV(32)
#undef MIMO_NV
  }
  return nullptr;
}

static int narrow_zi(int D) { return 16 * D <= 64 ? 1 : D <= 16 ? 4 : 8; }

// ---- the grouped (Dz-templated) variant: full feature map, Dz = 5 .. 32 — instantiated in mimo_narrow_grouped.hip (a translation
// unit of its own: the two halves of the instantiations compile side by side) ------------------------------------------------
narrow_fn pick_narrow_dt(int V, int D, int gibbs);
// Dz if the grouped variant serves (K, F, Dz), else 0 (MIMO_NARROW_GROUPED=0: off; MIMO_NARROW_GROUPED_MIN_D / _MAX_D: tuning knobs)
int narrow_dt(int K, int F, int D, int gibbs) {
  static const bool on = [] { const char* e = getenv("MIMO_NARROW_GROUPED"); return !e || atoi(e) != 0; }();
  static const int dmin = [] { const char* e = getenv("MIMO_NARROW_GROUPED_MIN_D"); return e ? atoi(e) : 5; }();
  static const int dmax = [] { const char* e = getenv("MIMO_NARROW_GROUPED_MAX_D"); return e ? atoi(e) : 32; }();
  if (!on || D < 5 || D > 32 || D < dmin || D > dmax || F != (D + 1) * (D + 2) / 2) return 0;
  const int V = narrow_v(K);
  if (!V || !pick_narrow_dt(V, D, gibbs)) return 0;
  // against the table-driven loops where both exist (profiles/r03_wide_sweep_grouped.txt, N = 2e6, us per pass, table / grouped):
  // softmax pass Dz=8 K=4 95 / 86, K=12 155 / 162, K=24 249 / 302; Dz=12 K=4 159 / 133, K=8 235 / 201, K=12 279 / 316; Dz=16 K=4
  // 258 / 205; Dz=24 K=8 1108 / 857; Dz=32 K=4 1552 / 840; Dz=5, 6: table-driven (a third more steps after the padding);
  // label pass: Dz=16 K=4 297 / 256, Dz=12 K=4 217 / 202, below that no difference
  static const bool always = [] { const char* e = getenv("MIMO_NARROW_GROUPED"); return e && atoi(e) == 2; }();
  const bool prefer = gibbs == 1 ? D >= 12 : (D >= 13 || (V <= 2 && D >= 7));
  if (!prefer && !always && pick_narrow(V, narrow_nsf(F), gibbs, narrow_zi(D))) return 0;
  return D;
}
int narrow_steps(int K, int F, int D, int gibbs) { return narrow_dt(K, F, D, gibbs) ? narrow_group_steps(D) : narrow_nsf(F); }
// position of feature (a, b), a <= b <= D, in the grouped order: step and index inside the step
void narrow_group_pos(int D, int a, int b, int* step, int* j) {
  int s0 = 0;
  for (int r = 0; r < a; ++r) s0 += (D + 1 - r + 3) / 4;
  *step = s0 + (b - a) / 4;
  *j = (b - a) % 4;
}

// Which (K, feature count F, Dz) the narrow kernels take (full structure or a reduced map alike: they read the feature
// table).  MIMO_NARROW=0 switches the route off, MIMO_NARROW_MIN_K / MIMO_NARROW_MAX_K move its K range (tuning knobs).
static int g_narrow_big_vi = 0;
void set_narrow_big_vi(int k) { g_narrow_big_vi = k; }

bool narrow_covers(int K, int F, int D, int ZS, int gibbs) {
  static const bool on = [] { const char* e = getenv("MIMO_NARROW"); return !e || atoi(e) != 0; }();
  static const int kmin_env = [] { const char* e = getenv("MIMO_NARROW_MIN_K"); return e ? atoi(e) : 0; }();
  // against the small-shape kernel (K <= 32 at Dz <= 2, K <= 16 at Dz = 3, 4) and the row-owner kernels behind it, N = 2e6, us per softmax
  // pass / sweep (profiles/r03_small_vs_narrow.txt): Dz=2 K=16 57 / 79 against 59 / 88, K=20 112 / 156 against 77 / 112, K=32 113 / 159 against 95 / 124;
  // Dz=3 K=12 86 / 146 against 61 / 88, K=24 171 / 166 against 87 / 130; Dz=4 K=8 55 / 81 against 59 / 91, K=16 110 / 163 against 83 / 105; Dz=1 K=32 83 / 126 against 73 / 110
  const int kmin = kmin_env > 0 ? kmin_env : D <= 2 ? 17 : D == 3 ? 9 : 12;
  static const int kmax = [] { const char* e = getenv("MIMO_NARROW_MAX_K"); return e ? atoi(e) : 256; }();
  // 129 .. 256 components over at most two contraction steps (mimo_narrow_big.hip), N = 2e6, ms per pass / sweep against the tile kernels
  // (profiles/r04_narrow_big.txt): one step (Dz = 1) K=144 0.33 / 0.27 against 0.76 / 0.43, K=256 0.70 / 0.43 against 1.14 / 0.62; two steps (Dz = 2)
  // K=160 0.43 / 0.33 against 0.82 / 0.43, K=192 0.79 / 0.39 against 0.92 / 0.52, K=224 1.28 / 0.45 against 1.25 / 0.56, K=256 3.48 / 0.50 against
  // 1.20 / 0.62 — the softmax pass of 52+ slots x 2 steps spills (64 l values + 128 accumulators + ~90 registers > 512): it stops at K = 192;
  // the label pass has no accumulators (two waves per SIMD, no scratch) and takes every K.  MIMO_NARROW_BIG_VI / _LABELS: tuning knobs
  // label pass (no accumulators: every slot count fits) over three steps (Dz = 3): 0.37 - 0.58 against 0.45 - 0.64 ms on the row-owner kernels,
  // K = 144 .. 256, N = 2e6 (profiles/r04_narrow_big.txt); four steps (Dz = 4) measured level with them (0.42 - 0.69 against 0.46 - 0.66) and stay there
  auto big_labels_steps = [] { static const int n = [] { const char* e = getenv("MIMO_NARROW_BIG_LABEL_STEPS"); return e ? atoi(e) : 3; }(); return n; };
  auto big_kmax = [](int g, int nsf) {
    static const int vi_env = [] { const char* e = getenv("MIMO_NARROW_BIG_VI"); return e ? atoi(e) : 0; }();
    const int vi = g_narrow_big_vi > 0 ? g_narrow_big_vi : vi_env;      // (mimo_tune "narrow_big_vi")
    static const int lab = [] { const char* e = getenv("MIMO_NARROW_BIG_LABELS"); return e ? atoi(e) : 256; }();
    return g == 1 ? lab : vi > 0 ? vi : nsf <= 1 ? 256 : 192;
  };
  static const bool wide_on = [] { const char* e = getenv("MIMO_NARROW_WIDE"); return !e || atoi(e) != 0; }();
  static const int wide_kmax = [] { const char* e = getenv("MIMO_NARROW_WIDE_MAX_K"); return e ? atoi(e) : 0; }();
  if (!on) return false;
  if (F > 16) {                 // few components, many features: the table-driven loops (Dz = 5 .. 16)
    // measured against the row-owner / tile kernels (tools/wide_sweep.py, N = 2e6, profiles/r03_wide_sweep.txt): K <= 16 wins wherever
    // the accumulators fit (Dz = 8: K = 4 276 -> 96 us, K = 16 271 -> 193; Dz = 16: K = 4 629 -> 256, K = 8 625 -> 349), K = 24 up
    // to Dz = 8 (248 against 274 us), K = 32 no longer (Dz = 8: 351 against 268)
    // Dz = 17 .. 32 (profiles/r03_wide_sweep_dz17_32.txt, against the two-stage tile kernels that pay for 16 components): K <= 4
    // Dz=17 1514 -> 290 us, Dz=24 2253 -> 538, Dz=32 3095 -> 1525 (one wave per SIMD from Dz = 26); K = 8 Dz=20 1633 -> 632, Dz=26 2304 -> 1219
    // 17 .. 24 features (NSF = 5, 6: the linear map of Dz = 16 .. 23 — tied covariances, the hierarchical drivers —, the full map of
    // Dz = 5): up to 64 components, 16 slots x 6 steps = 96 accumulators
    const int kmax_w = wide_kmax > 0 ? wide_kmax : (F <= 24 && D <= 16) ? 64 : (D <= 8 ? 24 : D <= 16 ? 16 : 8);
    if (!wide_on || K < 1 || K > kmax_w || D < 5 || D > 32 || (D > 16 && gibbs == 1)) return false;
  } else if (gibbs == 2) {
    return false;
  } else if (D > 4) {            // a reduced map of at most 16 features over wider rows
    if (!wide_on || D > 16 || K < 1 || K > 64) return false;
  } else if (K < kmin || K > kmax || K > 256 || (K > 128 && (narrow_nsf(F) > (gibbs == 1 ? big_labels_steps() : 2) || K > big_kmax(gibbs, narrow_nsf(F))))) {
    return false;
  }
  const int V = narrow_v(K);
  if (!V) return false;
  if (narrow_dt(K, F, D, gibbs)) return narrow_lds_bytes(V, narrow_group_steps(D), narrow_group_zs(D)) <= 64 * 1024;
  const int nsf = narrow_nsf(F);
  if (!pick_narrow(V, nsf, gibbs, narrow_zi(D))) return false;
  return narrow_lds_bytes(V, nsf, ZS) <= 64 * 1024;
}

int narrow_grid(const KernelArgs& a, int num_cu, int F, int gibbs) {
  const int dt = narrow_dt(a.K, F, a.D, gibbs);
  const int V = narrow_v(a.K), nsf = dt ? narrow_group_steps(dt) : narrow_nsf(F);
  int per_cu = narrow_waves(V, nsf, gibbs == 1 ? 1 : 0);
  if (narrow_fn fn = dt ? pick_narrow_dt(V, dt, gibbs) : pick_narrow(V, nsf, gibbs, narrow_zi(a.D))) {
    int nb = 0;
    const size_t lds = narrow_lds_bytes(V, nsf, dt ? narrow_group_zs(dt) : a.ZS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(fn), kNarrowWG, lds) == hipSuccess && nb > 0)
      per_cu = nb;
    (void)hipGetLastError();
  }
  if (const char* e = getenv("MIMO_NARROW_WG_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : per_cu;   // tuning knob
  const int64_t steps = (a.N + 15) / 16, need = (steps + 3) / 4;
  int64_t g = (int64_t)num_cu * per_cu;
  if (g > need) g = need;
  return (int)(g < 1 ? 1 : g);
}

hipError_t launch_narrow(const KernelArgs& a, int F, int gibbs, int grid, hipStream_t stream) {
  const int dt = narrow_dt(a.K, F, a.D, gibbs);
  const int V = narrow_v(a.K), nsf = dt ? narrow_group_steps(dt) : narrow_nsf(F);
  narrow_fn fn = dt ? pick_narrow_dt(V, dt, gibbs) : pick_narrow(V, nsf, gibbs, narrow_zi(a.D));
  if (!fn || 4 * V < a.K || 16 * a.D > 64 * narrow_zi(a.D)) return hipErrorInvalidValue;
  const size_t lds = narrow_lds_bytes(V, nsf, dt ? narrow_group_zs(dt) : a.ZS);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(kNarrowWG), lds, stream, a);
  return hipGetLastError();
}

}  // namespace mimo
