// gfx950 kernels for the NARROW shapes of the path: few features per component (F = (Dz+1)(Dz+2)/2 <= 16, i.e. Dz <= 4 —
// or a reduced feature map) with MANY components, 32 < K <= 128 — the shapes the reference's own ILR examples default to
// (examples/ilr/evaluate_sine.py:35, evaluate_chirp.py:38, evaluate_cmb.py:36, evaluate_step_poly.py:35: 50 experts over
// dx = dy = 1, i.e. Dz = 2; evaluate_sinc.py:35: 100) — and, with the same code, few components (K <= 16) at any Dz <= 16,
// K <= 8 up to Dz = 32 (softmax pass).
//
// Through the 16-padded v_mfma_f64_16x16x4 tiles such a pass pays for 16 features where 3 .. 15 exist (Dz = 2, K = 64:
// 20 % of the float64 peak).  Here both products run on v_mfma_f64_4x4x4_4b_f64 (4 blocks of 4 x 4 x 4; measured with
// tools/mfma_probe.hip: one per 16.2 cycles = the flop rate of 16x16x4; operand layout A: lane 16 k + 4 b + i,
// B: lane 16 k + 4 b + j, D: lane 16 i + 4 b + j), whose padding granule is 4 features and 4 components:
//
//   lane = (hi, b, lo) = (lane >> 4, (lane >> 2) & 3, lane & 3); a wave owns 16 rows per step, block b = rows 4 b .. 4 b + 3
//   L = Phi . Theta'   A = phi_{4 s + hi}(row 4 b + lo),  B = Theta[comp(c, lo)][4 s + hi] (LDS, the same for every b)
//                      -> lane (hi, b, lo) holds l of components comp(c, lo) = lo V + c, c < V, of row 4 b + hi
//   softmax / draw     in registers; the four lanes of a row are ADJACENT (lo): every cross-lane step is a DPP quad move
//   S += R' . Phi      A = the lane's r register of slot c AS IT IS (i = lo: component, k = hi: row),
//                      B = phi_{4 t + lo}(row 4 b + hi) / sum_k e  (the normaliser folded into the feature operand)
//                      -> lane (hi, b, lo) accumulates S[comp(c, hi)][4 t + lo] of block b's rows; V ceil(F / 4) accumulators
//
// No LDS round trip for l or r, no workgroup barrier in the loop; the four blocks' and the four waves' accumulators are
// added in a fixed order at the end (run-to-run bit-identical).  The label pass writes nothing but labels; their
// statistics come from label_stats_kernel (mimo_rowwave.hip).
//
// Reference behaviour reproduced: mimo/mixtures/gmm.py:62-75,227-259, ilr.py:66-84,161-194 (tables, softmax, draw),
// mimo/utils/stats.py:8-21 (label = #{k : u cum_K > cum_k}), gaussian.py:491-502, lingauss.py:306-322 (statistics).
// (kernel template and its compile-time helpers; instantiated by mimo_narrow.hip — slot loops and table-driven loops — and
//  mimo_narrow_grouped.hip — the Dz-templated grouped loops)
#pragma once
#include "mimo_device.h"
#include "mimo_extra.h"
#include "mimo_narrow_occ.h"

#include <cstdlib>

namespace mimo {

typedef void (*narrow_fn)(const KernelArgs);

constexpr int kNarrowWG = 256;      // 4 wavefronts; several workgroups per CU (registers decide)
#ifndef MIMO_NARROW_PF
#define MIMO_NARROW_PF 4
#endif
constexpr int kNarrowPF = MIMO_NARROW_PF;        // Theta slices in flight
#ifndef MIMO_NARROW_WHATIF
#define MIMO_NARROW_WHATIF 0     // diagnostic builds only: 1 no exp-table reads, 2 no Theta reads, 4 no Philox (results are wrong)
#endif
#ifndef MIMO_NARROW_EB
#define MIMO_NARROW_EB 8
#endif
#ifndef MIMO_NARROW_CHAINS
#define MIMO_NARROW_CHAINS 0     // 0: four chains for one slot over >= 64 steps, else one; 1, 2, 4: forced (tuning builds)
#endif
constexpr int kNarrowEB = MIMO_NARROW_EB;   // exponentials (table reads) in flight

namespace {

template <int CTRL>
__device__ __forceinline__ double quad_f64(double v) {       // DPP quad_perm on both halves
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int quad_i32(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
constexpr int kQuadXor1 = 0xB1;     // [1, 0, 3, 2]
constexpr int kQuadXor2 = 0x4E;     // [2, 3, 0, 1]
constexpr int kQuadUp1 = 0x90;      // [0, 0, 1, 2]: lane lo reads lo - 1
constexpr int kQuadUp2 = 0x40;      // [0, 0, 0, 1]: lane lo reads lo - 2
constexpr int kQuadLast = 0xFF;     // [3, 3, 3, 3]

}  // namespace

// Waves per SIMD every instantiation is compiled for: the largest count at which hipcc allocates with at most 48 bytes of scratch
// (mimo_narrow_occ.h, written by tools/narrow_occupancy.py from -Rpass-analysis=kernel-resource-usage of builds forced to
// 4, 3, 2 and 1 waves).  Registers: V l values + V NSF accumulators (doubles) + 70 .. 100 besides; a bound that is too tight
// spills and costs far more than the lost wave: V = 16, NSF = 2 forced to three waves (168 registers, 220 bytes of scratch)
// ran 285 us where the two-wave build runs 152 us (N = 2e6).
#ifndef MIMO_NARROW_BIG_LABEL_WAVES
#define MIMO_NARROW_BIG_LABEL_WAVES 2
#endif
#ifdef MIMO_NARROW_FORCE_WAVES
constexpr int narrow_waves(int, int, int) { return MIMO_NARROW_FORCE_WAVES; }
#else
// (the table covers NSF <= 4 — its key does not separate larger NSF; the table-driven loops take what the allocator gives them)
// (V > 32 — 129 .. 256 components over one or two contraction steps, Dz <= 2: mimo_narrow_big.hip — is not in the table: V l values
// and V NSF accumulators take the whole unified register file of one wave per SIMD in the softmax pass; the label pass has no accumulators)
constexpr int narrow_waves(int V, int NSF, int MODE) { return NSF > 4 ? 1 : V > 32 ? (MODE == 1 ? MIMO_NARROW_BIG_LABEL_WAVES : 1) : narrow_occ(V, NSF, MODE) & 7; }
#endif
// ... and which of the two loop bodies: "lean" (bit 3 of the table entry) keeps the exponentials inline and reads the operand
// factors at the top of a step — fewer live registers, what the large slot counts need; the other one batches the table reads
// and prefetches the next step's factors
#ifdef MIMO_NARROW_FORCE_LEAN
constexpr bool narrow_lean(int, int, int) { return MIMO_NARROW_FORCE_LEAN != 0; }
#else
constexpr bool narrow_lean(int V, int NSF, int MODE) { return NSF > 4 || V > 32 || (narrow_occ(V, NSF, MODE) & 8) != 0; }
#endif

// The GROUPED feature order of the Dz-templated variant (DT = Dz >= 5, full map): row a of the upper triangle of z~ z~' —
// the pairs (a, a), (a, a + 1), .., (a, Dz) — padded to whole steps of four, so that a step is (a, b0) and the lane's feature of
// it is z~[a] z~[b0 + j], j = its index inside the step.  A lane then keeps TWO register copies of its row — z~[i] and the copy
// shifted by j, z~[i + j] (slots Dz + 1 .. Dz + 3 of the row are zero) — and every operand of both products is one
// register-register product with compile-time indices: 2 (Dz + 2) LDS reads per product and 16-row step instead of the ~2.5 per
// matrix instruction of the table-driven loops (which are bound by exactly that: LDS bandwidth, 56 - 64 cycles per step at V = 1).
template <int D>
struct NarrowGroup {
  static constexpr int nst() { int n = 0; for (int r = 0; r <= D; ++r) n += (D + 1 - r + 3) / 4; return n; }
  int a[nst() > 0 ? nst() : 1], b0[nst() > 0 ? nst() : 1];
  constexpr NarrowGroup() : a{}, b0{} {
    int s = 0;
    for (int r = 0; r <= D; ++r)
      for (int b = r; b <= D; b += 4) { a[s] = r; b0[s] = b; ++s; }
  }
};
constexpr int narrow_group_steps(int D) { int n = 0; for (int r = 0; r <= D; ++r) n += (D + 1 - r + 3) / 4; return n; }
constexpr int narrow_group_zs(int D) { return (D + 4) | 1; }      // row stride of the grouped variant: z, 1, three zero slots

// MODE 0: softmax + statistics (fast mean-field / EM pass), 1: label draw, 2: label draw + the statistics of the labels in the same
// pass (the one-hot row of the drawn label takes the place of the responsibilities in the second product: a sweep costs what a
// softmax pass costs and reads Z once — for the table-driven / grouped loops, whose second product is cheap next to a second pass)
template <int V, int NSF, int MODE, int ZI, int DT = 0>
__global__ __launch_bounds__(kNarrowWG, narrow_waves(V, NSF, MODE)) void narrow_kernel(const KernelArgs a) {
  constexpr bool STATS = MODE != 1;
  static_assert(MODE != 2 || NSF > 4, "label draw + statistics: table-driven and grouped loops only");
  static_assert(DT == 0 || (DT >= 5 && NSF == narrow_group_steps(DT)), "grouped variant: NSF = steps of the grouped order");
  constexpr NarrowGroup<DT> GR{};
  constexpr bool LEAN = narrow_lean(V, NSF, MODE);
  // NSF > 4 (more than 16 features: Dz >= 5, few components): the operand factors of a step come from a small LDS table of
  // packed byte offsets inside the step loops instead of 4 NSF address registers and 2 NSF operand registers
  constexpr bool FT = NSF > 4;
  constexpr int NP = FT ? 1 : NSF;
  constexpr int EB = LEAN ? 1 : kNarrowEB;                    // exponentials per batch (1: inline, hipcc schedules them)
  extern __shared__ __align__(16) unsigned char smem[];
  const int ZS = DT ? narrow_group_zs(DT) : a.ZS;
  double* Th = reinterpret_cast<double*>(smem);               // [NSF V + PF][16]
  double* etab = Th + (size_t)(NSF * V + kNarrowPF) * 16;     // [kExpTab]; the epilogue's scratch aliases it
  double* Zall = etab + kExpTab;                              // [4 waves][16][ZS]
  double* sred = Zall + (size_t)4 * 16 * ZS;                  // [4]
  uint32_t* ftab = reinterpret_cast<uint32_t*>(sred + 4);     // [NSF][4] (FT only)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 4, b = (lane >> 2) & 3, lo = lane & 3;
  const int D = a.D, K = a.K;
  const int64_t N = a.N;
  double* Zw = Zall + (size_t)wave * 16 * ZS;

  for (int e = tid; e < NSF * V * 16; e += kNarrowWG) Th[e] = a.theta[e];
  for (int e = tid; e < kNarrowPF * 16; e += kNarrowWG) Th[NSF * V * 16 + e] = 0.0;
  for (int e = tid; e < kExpTab; e += kNarrowWG) etab[e] = exp_tab_entry_c(e);
  if constexpr (FT && DT == 0)
    for (int e = tid; e < 4 * NSF; e += kNarrowWG) ftab[e] = 8u * a.feat[2 * e] | (8u * a.feat[2 * e + 1]) << 16;
  if constexpr (DT > 0)                        // the zero slots behind [z, 1] (slot Dz + 1 is rewritten with every step)
    if (tid < 64) { Zall[(size_t)tid * ZS + DT + 2] = 0.0; Zall[(size_t)tid * ZS + DT + 3] = 0.0; }
  // label pass with a.fuse_hist: the histogram of the labels drawn here, for the slot table of label_stats_slots_kernel behind it
  // (as gibbs_rowwave_kernel: no label_hist_kernel pass over the labels)
  __shared__ uint32_t hloc[MODE == 1 ? 256 : 1];
  if constexpr (MODE == 1) hloc[tid] = 0u;
  wg_sync();

  const int64_t nsteps = (N + 15) / 16;
  const int64_t nwaves = (int64_t)gridDim.x * 4, wv = (int64_t)blockIdx.x * 4 + wave;
  // z rows of a 16-row step: element e = lane + 64 i of the (16, D) block
  int zoff[ZI];
#pragma unroll
  for (int i = 0; i < ZI; ++i) {
    const int e = lane + 64 * i, r = e / D;
    zoff[i] = e < 16 * D ? r * ZS + (e - r * D) : -1;
  }
  double zr[ZI];
  auto load_z = [&](int64_t t) {
    const int64_t base = t * 16 * D, total = N * D;
#pragma unroll
    for (int i = 0; i < ZI; ++i) {
      const int64_t gidx = base + lane + 64 * i;
      zr[i] = (zoff[i] >= 0 && gidx < total) ? a.Z[gidx] : 0.0;
    }
  };
  if (wv < nsteps) load_z(wv);

  // operand addresses, fixed for the whole kernel: first product feature 4 s + hi of row 4 b + lo, second product
  // feature 4 s + lo of row 4 b + hi (z~ = [z, 1, 0]: padded features read the zero slot)
  const double* row0 = Zw + (4 * b + lo) * ZS;
  const double* row1 = Zw + (4 * b + hi) * ZS;
  const double* ea[NP];
  const double* eb[NP];
  const double* sa[MODE == 0 ? NP : 1];
  const double* sb[MODE == 0 ? NP : 1];
  const char* z0b = reinterpret_cast<const char*>(row0);
  const char* z1b = reinterpret_cast<const char*>(row1);
  auto feat0 = [&](int s) -> double {          // FT: feature 4 s + hi of row 4 b + lo
    const uint32_t u = ftab[4 * s + hi];
    return *reinterpret_cast<const double*>(z0b + (u & 0xffffu)) * *reinterpret_cast<const double*>(z0b + (u >> 16));
  };
  auto feat1 = [&](int s) -> double {          // FT: feature 4 s + lo of row 4 b + hi
    const uint32_t u = ftab[4 * s + lo];
    return *reinterpret_cast<const double*>(z1b + (u & 0xffffu)) * *reinterpret_cast<const double*>(z1b + (u >> 16));
  };
#pragma unroll
  for (int s = 0; s < NP; ++s) {
    if constexpr (FT) break;
    ea[s] = row0 + a.feat[2 * (4 * s + hi)];
    eb[s] = row0 + a.feat[2 * (4 * s + hi) + 1];
    if constexpr (MODE == 0) {
      sa[s] = row1 + a.feat[2 * (4 * s + lo)];
      sb[s] = row1 + a.feat[2 * (4 * s + lo) + 1];
    }
  }
  const double* thl = Th + 4 * hi + lo;

  double sacc[STATS ? V : 1][STATS ? NSF : 1];
  if constexpr (STATS) {
#pragma unroll
    for (int c = 0; c < V; ++c)
#pragma unroll
      for (int s = 0; s < NSF; ++s) sacc[c][s] = 0.0;
  }
  double sc_lse = 0.0, sc_prod = 1.0;
  int since_flush = 0;
  double ubatch = 0.0;
  int uphase = 0;

  // The z~ rows of step t + nwaves are staged (registers -> the wave's LDS block) at the top of step t, and their operand
  // factors are read back between the softmax and the second product of step t: no LDS round trip lies open in the loop.
  auto stage_z = [&](int64_t t) {
#pragma unroll
    for (int i = 0; i < ZI; ++i)
      if (zoff[i] >= 0) Zw[zoff[i]] = zr[i];
    if (lane < 16) {
      Zw[lane * ZS + D] = (t * 16 + lane) < N ? 1.0 : 0.0;     // rows past N: every feature 0 — nothing reaches the statistics
      Zw[lane * ZS + D + 1] = 0.0;
    }
  };
  double fa[NP], fb[NP], ga[MODE == 0 ? NP : 1], gb[MODE == 0 ? NP : 1];      // operand factors of the step in hand
  auto read_factors = [&]() {
    if constexpr (!FT) {
#pragma unroll
      for (int s = 0; s < NP; ++s) { fa[s] = *ea[s]; fb[s] = *eb[s]; }
      if constexpr (MODE == 0) {
#pragma unroll
        for (int s = 0; s < NP; ++s) { ga[s] = *sa[s]; gb[s] = *sb[s]; }
      }
    }
  };
  if (!LEAN && wv < nsteps) {
    stage_z(wv);
    if (wv + nwaves < nsteps) load_z(wv + nwaves);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    read_factors();
  }

  // ---- S += R' . Phi: rr = the lane's r values (the A operand as it is), inv = 1 / sum e (1 for one-hot rows) ----------------
  auto stats_product = [&](double (&rr)[V], double (&bv)[MODE == 0 ? NP : 1], double inv) {
    if constexpr (STATS) {
      double bq = 0.0;
      if constexpr (FT && DT == 0) bq = feat1(0) * inv;
      double zB[DT + 1], zT[DT + 1];                       // grouped: row 4 b + hi and its copy shifted by lo
      if constexpr (DT > 0) {
        const double* r1l = row1 + lo;
#pragma unroll
        for (int i = 0; i <= DT; ++i) { zB[i] = row1[i]; zT[i] = r1l[i]; }
        if constexpr (MODE == 0) {
#pragma unroll
          for (int c = 0; c < V; ++c) rr[c] *= inv;          // (V products instead of one per feature step)
        }
      }
#pragma unroll
      for (int s = 0; s < NSF; ++s) {
        double bcur;
        if constexpr (DT > 0) {
          bcur = zB[GR.a[s]] * zT[GR.b0[s]];
        } else if constexpr (FT) {
          bcur = bq;
          if (s + 1 < NSF) bq = MODE == 0 ? feat1(s + 1) * inv : feat1(s + 1);
        } else {
          bcur = bv[s];
        }
#pragma unroll
        for (int c = 0; c < V; ++c)
          sacc[c][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(rr[c], bcur, sacc[c][s], 0, 0, 0);
      }
    }
  };

  for (int64_t t = wv; t < nsteps; t += nwaves) {
    const int64_t n1 = t * 16 + 4 * b + hi;           // the row this lane normalises / draws for
    const bool valid = n1 < N;
    double av[NP], bv[MODE == 0 ? NP : 1];
    if constexpr (LEAN) {                               // stage this step's rows and read its factors right away
      stage_z(t);
      if (t + nwaves < nsteps) load_z(t + nwaves);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      read_factors();
    }
    if constexpr (!FT) {
#pragma unroll
      for (int s = 0; s < NP; ++s) av[s] = fa[s] * fb[s];
      if constexpr (MODE == 0) {
#pragma unroll
        for (int s = 0; s < NP; ++s) bv[s] = ga[s] * gb[s];
      }
    }
    // ---- the next step's rows go to the LDS block now (this step's factors are in registers)
    const bool more = !LEAN && t + nwaves < nsteps;
    if (more) {
      stage_z(t + nwaves);
      if (t + 2 * nwaves < nsteps) load_z(t + 2 * nwaves);
    }

    // ---- L = Phi . Theta': slice e = s V + c of the operand image -------------------------------------------
    double acc[V];
#pragma unroll
    for (int c = 0; c < V; ++c) acc[c] = 0.0;
    // one slot and many steps: its products would form ONE dependent chain of NSF matrix instructions — split in four (measured,
    // N = 2e6, one / four chains: Dz=20 409 / 387 us, Dz=32 841 / 771 us; shorter chains and two slots: no difference or slower)
    constexpr int NCHAIN = !FT ? 1 : MIMO_NARROW_CHAINS > 0 ? MIMO_NARROW_CHAINS : (V == 1 && NSF >= 64) ? 4 : 1;
    double accx[V][NCHAIN > 1 ? NCHAIN - 1 : 1];
    if constexpr (NCHAIN > 1) {
#pragma unroll
      for (int c = 0; c < V; ++c)
#pragma unroll
        for (int p = 0; p < NCHAIN - 1; ++p) accx[c][p] = 0.0;
    }
    {
      double ring[kNarrowPF];
#pragma unroll
      for (int e = 0; e < kNarrowPF; ++e) ring[e] = thl[e * 16];
      double aq = 0.0;
      if constexpr (FT && DT == 0) aq = feat0(0);
      double zA[DT + 1], zS[DT + 1];                       // grouped: the lane's row and its copy shifted by hi
      if constexpr (DT > 0) {
        const double* r0h = row0 + hi;
#pragma unroll
        for (int i = 0; i <= DT; ++i) { zA[i] = row0[i]; zS[i] = r0h[i]; }
      }
#pragma unroll
      for (int s = 0; s < NSF; ++s) {
        double acur;
        if constexpr (DT > 0) {
          acur = zA[GR.a[s]] * zS[GR.b0[s]];
        } else if constexpr (FT) {
          acur = aq;
          if (s + 1 < NSF) aq = feat0(s + 1);        // (one step ahead: the LDS round trip hides under this step's products)
        } else {
          acur = av[s];
        }
#pragma unroll
        for (int c = 0; c < V; ++c) {
          const int e = s * V + c;
          const double tv = ring[e % kNarrowPF];
          if (!(MIMO_NARROW_WHATIF & 2)) ring[e % kNarrowPF] = thl[(e + kNarrowPF) * 16];     // (the last reads take the zero slices behind the image)
          if (NCHAIN > 1 && s % NCHAIN != 0)
            accx[c][s % NCHAIN - 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(acur, tv, accx[c][s % NCHAIN - 1], 0, 0, 0);
          else
            acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(acur, tv, acc[c], 0, 0, 0);
        }
      }
    }
    if constexpr (NCHAIN > 1) {
#pragma unroll
      for (int c = 0; c < V; ++c) {
        if constexpr (NCHAIN == 2) acc[c] += accx[c][0];
        else if constexpr (NCHAIN == 4) acc[c] = (acc[c] + accx[c][0]) + (accx[c][1] + accx[c][2]);
      }
    }

    // ---- this lane: components lo V .. lo V + V - 1 of row 4 b + hi; the row's other quarters sit in the adjacent lanes
    __builtin_amdgcn_s_setprio(2);
    double m;
    {
      double mv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) mv[i] = acc[i < V ? i : 0];
#pragma unroll
      for (int c = 4; c < V; ++c) mv[c & 3] = fmax(mv[c & 3], acc[c]);
      m = fmax(fmax(mv[0], mv[1]), fmax(mv[2], mv[3]));
      m = fmax(m, quad_f64<kQuadXor1>(m));
      m = fmax(m, quad_f64<kQuadXor2>(m));
    }
    // e = exp(l - max) in batches of EB: all table reads of a batch are in flight before the first is used
    if constexpr (MODE == 0) {
      double sv[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int c0 = 0; c0 < V; c0 += EB) {
        double tv[EB];
        int nn[EB];
#pragma unroll
        for (int i = 0; i < EB; ++i)
          if (c0 + i < V) {
            double r;
            exp_c_issue(acc[c0 + i] - m, r, nn[i]);
            acc[c0 + i] = r;
            tv[i] = (MIMO_NARROW_WHATIF & 1) ? __hiloint2double(0x3ff00000 - ((nn[i] & 2047) << 9), nn[i]) : etab[nn[i] & (kExpTab - 1)];
          }
        if (!LEAN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < EB; ++i)
          if (c0 + i < V) {
            acc[c0 + i] = exp_c_finish(acc[c0 + i], nn[i], tv[i]);
            sv[(c0 + i) & 3] += acc[c0 + i];
          }
        if (!LEAN) __builtin_amdgcn_sched_barrier(0);
      }
      double ssum = (sv[0] + sv[1]) + (sv[2] + sv[3]);
      ssum += quad_f64<kQuadXor1>(ssum);
      ssum += quad_f64<kQuadXor2>(ssum);
      double inv = __builtin_amdgcn_rcp(ssum);
      inv = fma(fma(-ssum, inv, 1.0), inv, inv);
      inv = fma(fma(-ssum, inv, 1.0), inv, inv);
      if (lo == 0 && valid) { sc_lse += m; sc_prod *= ssum; }
      // per-row weights of the statistics (mimo_estep_weighted: the outer responsibilities of a mixture of mixtures; the NaN-row
      // mask): they ride on the normaliser — tables and scalars stay unweighted (hgmm.py:199-207)
      if (a.u) inv *= valid ? a.u[n1] : 0.0;
      if constexpr (!FT) {
#pragma unroll
        for (int s = 0; s < NP; ++s) bv[s] *= inv;      // r = e / sum e: the normaliser rides on the feature operand
      }
      __builtin_amdgcn_s_setprio(0);
      if (more) {                                       // the next step's operand factors: in flight under the second product
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        read_factors();
      }
      stats_product(acc, bv, inv);
      if (++since_flush == 64) {          // (4 V)^64 <= 128^64 = 2^448 stays inside the float64 range
        sc_lse += log(sc_prod);
        sc_prod = 1.0;
        since_flush = 0;
      }
    } else {
      // inclusive cumulative sums inside chunks of 8 (independent chains across the chunks); the draw is the inverse CDF
      // on the UNNORMALISED sums (mimo/utils/stats.py:10-17: label = #{k : u cum_K > cum_k})
      constexpr int NCH = (V + 7) / 8;
      static_assert(EB == 8 || EB == 4 || EB == 1, "a batch of exponentials divides a chunk of the cumulative sums");
      double base[NCH + 1];
      base[0] = 0.0;
      if (more) {                                       // the next step's operand factors: in flight under the draw
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        read_factors();
      }
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const int len = V - 8 * ch < 8 ? V - 8 * ch : 8;
#pragma unroll
        for (int h0 = 0; h0 < 8; h0 += EB) {
          double tv[EB];
          int nn[EB];
#pragma unroll
          for (int i = 0; i < EB; ++i)
            if (h0 + i < len) {
              double r;
              exp_c_issue(acc[8 * ch + h0 + i] - m, r, nn[i]);
              acc[8 * ch + h0 + i] = r;
              tv[i] = (MIMO_NARROW_WHATIF & 1) ? __hiloint2double(0x3ff00000 - ((nn[i] & 2047) << 9), nn[i]) : etab[nn[i] & (kExpTab - 1)];
            }
          if (!LEAN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < EB; ++i)
            if (h0 + i < len) {
              const double ev = exp_c_finish(acc[8 * ch + h0 + i], nn[i], tv[i]);
              acc[8 * ch + h0 + i] = (h0 + i > 0) ? acc[8 * ch + h0 + i - 1] + ev : ev;
            }
          if (!LEAN) __builtin_amdgcn_sched_barrier(0);
        }
        base[ch + 1] = acc[8 * ch + len - 1];
        if (LEAN) __builtin_amdgcn_sched_barrier(0);       // one chunk of exp chains in flight at a time (register pressure)
      }
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) base[ch + 1] += base[ch];
      const double cum = base[NCH];
      double incl = cum;                      // inclusive scan over the four quarters of the row (adjacent lanes)
      {
        double v = quad_f64<kQuadUp1>(incl);  if (lo >= 1) incl += v;
        v = quad_f64<kQuadUp2>(incl);         if (lo >= 2) incl += v;
      }
      double excl = quad_f64<kQuadUp1>(incl);
      if (lo == 0) excl = 0.0;
      const double ctot = quad_f64<kQuadLast>(incl);
      double uu;
      if (a.u) {
        uu = valid ? a.u[n1] : 0.0;
      } else {
        // Philox uniforms four steps at a time: lane (hi, b, lo) draws the uniform of row 4 b + hi of this wave's step
        // t + lo nwaves — the same counters as one draw per step, the same labels
        if (uphase == 0 && !(MIMO_NARROW_WHATIF & 4))
          ubatch = philox_uniform(a.seed, (uint64_t)(a.row0 + (t + (int64_t)lo * nwaves) * 16 + 4 * b + hi), a.sweep);
        uu = __shfl(ubatch, (lane & ~3) | uphase);
        uphase = (uphase + 1) & 3;
      }
      const double tl = uu * ctot - excl;
      int cnt = 0;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const double tc = tl - base[ch];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (8 * ch + i < V) cnt += tc > acc[8 * ch + i] ? 1 : 0;
      }
      cnt += quad_i32<kQuadXor1>(cnt);
      cnt += quad_i32<kQuadXor2>(cnt);
      const int label = cnt < K ? cnt : K - 1;
      if (lo == 0 && valid) {
        a.labels[n1] = label;
        if constexpr (MODE == 1)
          if (a.fuse_hist) atomicAdd(&hloc[label], 1u);
      }
      __builtin_amdgcn_s_setprio(0);
      if constexpr (MODE == 2) {             // the one-hot row of the label (rows past N: every feature is zero)
#pragma unroll
        for (int c = 0; c < V; ++c) acc[c] = (lo * V + c == label) ? 1.0 : 0.0;
        double nobv[1] = {0.0};
        stats_product(acc, nobv, 1.0);
      }
    }
  }

  if constexpr (MODE == 1) {
    if (a.fuse_hist) {                       // (uniform)
      wg_sync();
      if (hloc[tid]) atomicAdd(&a.aux[tid], hloc[tid]);
    }
  }
  if constexpr (STATS) {
    // ---- per-workgroup partial block: the four blocks and the four waves added in a fixed order, GB accumulators at a time
    const int FT = a.F16_total, Kpad = a.K16 * 16;
    const size_t pstride = (size_t)Kpad * FT + 4;
    double* P = a.partials + (size_t)blockIdx.x * pstride;
    double* red = etab;                                  // [GB][4 waves][64] = 16 KB: the (now idle) exp table
    constexpr int GB = 8, NACC = V * NSF;
    // rows / columns of the block this kernel has no accumulator for
    for (int e = tid; e < Kpad * FT; e += kNarrowWG) {
      const int k = e / FT, f = e - k * FT;
      if (k >= 4 * V || f >= (DT ? (DT + 1) * (DT + 2) / 2 : 4 * NSF)) P[e] = 0.0;
    }
#pragma unroll
    for (int g0 = 0; g0 < NACC; g0 += GB) {
      wg_sync();
#pragma unroll
      for (int i = 0; i < GB; ++i)
        if (g0 + i < NACC) red[(i * 4 + wave) * 64 + lane] = sacc[(g0 + i) / NSF][(g0 + i) % NSF];
      wg_sync();
      if (tid < 16 * GB) {
        const int i = tid >> 4, ci = (tid >> 2) & 3, fj = tid & 3, idx = g0 + i;
        if (idx < NACC) {
          const int c = idx / NSF, s = idx - c * NSF;
          double tot = 0.0;
#pragma unroll
          for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) tot += red[(i * 4 + w) * 64 + 16 * ci + 4 * bb + fj];
          const int k = ci * V + c;
          int f = 4 * s + fj;
          if constexpr (DT > 0) {            // step s of the grouped order = (row r, first column b0): feature (r, b0 + fj) or padding
            int r = 0, s0 = 0;
            while (s0 + (DT + 1 - r + 3) / 4 <= s) { s0 += (DT + 1 - r + 3) / 4; ++r; }
            const int col = r + 4 * (s - s0) + fj;
            f = col <= DT ? r * (DT + 1) - r * (r - 1) / 2 + (col - r) : FT;
          }
          if (k < Kpad && f < FT) P[(size_t)k * FT + f] = k < K ? tot : 0.0;
        }
      }
    }
    sc_lse += log(sc_prod);
    sc_lse = wave_sum(sc_lse);
    wg_sync();
    if (lane == 0) sred[wave] = sc_lse;
    wg_sync();
    if (tid == 0 && a.write_scalars) {
      double* Ps = P + (size_t)Kpad * FT;
      Ps[0] = (sred[0] + sred[1]) + (sred[2] + sred[3]); Ps[1] = 0.0; Ps[2] = 0.0; Ps[3] = 0.0;
    }
  }
}

// ------------------------------------------------------------------------------------------
// which instantiations exist (shared by the translation units that instantiate them: mimo_narrow.hip — up to four steps —,
// mimo_narrow_table.hip — five and more)
// ------------------------------------------------------------------------------------------
constexpr int kNarrowMaxAcc = 96, kNarrowMaxVWide = 8, kNarrowMaxAccXWide = 200;
template <int V, int NSF>
static narrow_fn pick_narrow_mode(int gibbs, int zi) {
  if constexpr (NSF <= 4) {
    if (zi == 1) return gibbs == 2 ? nullptr : gibbs ? narrow_kernel<V, NSF, 1, 1> : narrow_kernel<V, NSF, 0, 1>;
    // up to 16 features of rows with Dz = 5 .. 16 (the linear map of tied covariances / the hierarchical drivers at Dz <= 15, the
    // diagonal map up to Dz = 7): the same loops over 4-element row loads, up to 64 components
    if constexpr (V <= 16 && NSF >= 2)
      if (zi == 4) return gibbs == 2 ? nullptr : gibbs ? narrow_kernel<V, NSF, 1, 4> : narrow_kernel<V, NSF, 0, 4>;
  } else if constexpr (NSF <= 39 && V <= (NSF <= 6 ? 16 : kNarrowMaxVWide) && V * NSF <= kNarrowMaxAcc) {
    if (zi == 4) return gibbs == 2 ? narrow_kernel<V, NSF, 2, 4> : gibbs ? narrow_kernel<V, NSF, 1, 4> : narrow_kernel<V, NSF, 0, 4>;
  } else if constexpr (NSF > 39 && V <= 2 && V * NSF <= kNarrowMaxAccXWide) {
    // Dz = 17 .. 32 (ZI = 8), softmax + statistics pass or label draw + statistics with K <= 8: one wave per SIMD, the accumulators
    // take the second half of the unified register file
    if (zi == 8 && gibbs != 1) return gibbs == 2 ? narrow_kernel<V, NSF, 2, 8> : narrow_kernel<V, NSF, 0, 8>;
  }
  return nullptr;
}

}  // namespace mimo
