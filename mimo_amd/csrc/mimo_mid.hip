// gfx950 kernel for the MID shapes of the softmax + statistics pass: a handful to a few dozen components over WIDE rows (K <= 32 up to
// Dz = 32, F = (Dz+1)(Dz+2)/2 up to 561 features; K <= 48 up to Dz = 26, K <= 64 up to Dz = 21, K <= 96 up to Dz = 14 — while the
// accumulators fit), and the component counts between the tile kernels' sizes (65 .. 96: five or six row blocks instead of eight).  The reference's hierarchical / tied examples with few components over wide
// inputs land here as D grows (examples/hgmm/*.py, examples/tgmm/vi_tgmm.py; mimo/mixtures/gmm.py:244-259 E-step,
// gaussian.py:491-502 statistics).  Until round 4 they ran the round-1 two-stage pair (estep_chunked_kernel, which pays for 64
// component slots whatever K is, + fused_kernel statistics per column group through the (K, N) table in HBM): 0.13 - 0.50 of the
// float64 rate.  The narrow kernels (4x4x4 tiles, mimo_narrow_kernel.h) run out of accumulators: their layout keeps one partial
// block per 4-row block, K F x 4 doubles.
//
// One pass, two ownerships (v_mfma_f64_16x16x4_f64 for both products):
//   E-step: a wave owns 16 ROWS and all features.   L (16 KB comps x 16 rows) = Theta . Phi'
//       A = Theta slice (16 comps x 4 features) streamed from the L2-resident image through a register ring,
//       B = phi_{step s, q}(row j) = z~[a_s] z~[b0_s + q]: the GROUPED feature order of the narrow kernels (rows of the upper
//           triangle padded to steps of four), so that with the lane's row shifted by q in registers every operand is ONE
//           register-register product — no LDS access per matrix instruction at all;
//       softmax over the row's 16 KB components in registers (4 KB per lane, the row's four lanes 16 apart); r -> LDS.
//   statistics: a wave owns feature COLUMN BLOCKS w, w + NW, .. and all 16 NW rows of the super-step.   S += R . Phi
//       A = r of four rows from the shared R tile (one read per four rows, reused by every column block),
//       B = phi_{16 cb + j}(row 4 t + q): two reads of the shared z tile + one product, reused by the KB row blocks;
//       the accumulators of a wave are ITS columns of the block: no cross-wave reduction, K F doubles per workgroup in all.
// z and R tiles are double-buffered: ONE workgroup barrier per super-step of 16 NW rows.  Fixed summation order: bit-identical
// from run to run.  Per-row weights (mimo_estep_weighted, the NaN-row mask) ride on the normaliser.
#include "mimo_narrow_kernel.h"      // NarrowGroup, narrow_group_steps, narrow_group_pos

namespace mimo {

constexpr int kMidPF = 8;                                        // Theta slices in flight per wave
constexpr int mid_zs(int D) { return (D + 4) | 1; }              // row stride: z, 1, three zero slots; odd
constexpr int mid_ncb(int D) { return ((D + 1) * (D + 2) / 2 + 15) / 16; }
// Registers of an instantiation (calibrated on the compiler's reports: D=24 KB=1 NW=4 216, D=20 KB=2 NW=4 236, D=16 KB=3 NW=4 240):
// the wave's accumulators — KB x its column blocks x 8 —, the E-step's L tile (8 KB), the lane's row shifted by q (2 (Dz + 1)),
// ~110 besides (Theta ring, prefetched rows, operand addresses and factors, the softmax).
constexpr int mid_regs(int D, int KB, int NW) { return 8 * KB * ((mid_ncb(D) + NW - 1) / NW) + 8 * KB + 2 * (D + 1) + 110; }
// waves per workgroup: 4 (two workgroups per CU) while that fits 256 registers, else 8 (one workgroup per CU); up to ~50 bytes of
// scratch outside the loops cost nothing measurable (Dz=26 K=48: 0.80 of the float64 rate with 52 bytes)
constexpr int mid_nw(int D, int KB) { return mid_regs(D, KB, 4) <= 250 ? 4 : 8; }
constexpr bool mid_fits(int D, int KB) { return mid_regs(D, KB, mid_nw(D, KB)) <= (KB <= 2 ? 275 : 264); }
// ... and beyond that ONE wave per SIMD (a four-wave workgroup alone on its CU): the wave has the whole unified register file, 512
// registers, the accumulators spill over into its second half; Theta slices 16 deep in flight (nothing else hides the L2 latency)
constexpr bool mid_big(int D, int KB) { return !mid_fits(D, KB) && mid_regs(D, KB, 4) + 16 <= 470; }
constexpr bool mid_exists(int D, int KB) { return KB >= 1 && KB <= 8 && D >= 5 && D <= 32 && (mid_fits(D, KB) || mid_big(D, KB)); }
constexpr int mid_waves(int D, int KB) { return mid_big(D, KB) ? 4 : mid_nw(D, KB); }
// the z and R tiles are double-buffered (one barrier per super-step) where two copies fit, else single (two barriers)
constexpr size_t mid_tile_bytes(int D, int KB) { return sizeof(double) * (size_t)16 * mid_waves(D, KB) * (mid_zs(D) + 16 * KB + 1); }
constexpr int mid_wgs_per_cu(int D, int KB) { return mid_big(D, KB) ? 1 : mid_nw(D, KB) == 4 ? 2 : 1; }
constexpr int mid_nbuf(int D, int KB) { return 2 * mid_tile_bytes(D, KB) * mid_wgs_per_cu(D, KB) + 2048 <= 160 * 1024 ? 2 : 1; }
constexpr size_t mid_lds_bytes(int D, int KB) { return mid_nbuf(D, KB) * mid_tile_bytes(D, KB) + sizeof(double) * (64 + 8); }

// MODE 0: softmax + statistics.  MODE 1: label draw (Gibbs sweep of the same shapes): the E-step as above, then the inverse-CDF draw on
// the unnormalised cumulative sums in registers (mimo/utils/stats.py:8-21) — the operand image is permuted so that lane (q, j) holds
// the CONTIGUOUS quarter q V .. q V + V - 1 (V = 4 KB) of row j's components (upload_theta_mid, as the row-owner label kernels);
// no R tile, no second product, no workgroup barrier: the waves run on their own, the statistics of the labels come from the
// label-statistics kernels behind it.
template <int DT, int KB, int NW, int MODE = 0>
__global__ __launch_bounds__(64 * NW, MODE == 1 ? 2 : mid_wgs_per_cu(DT, KB)) void mid_kernel(const KernelArgs a) {
  constexpr int PF = (MODE == 0 && mid_big(DT, KB)) ? 16 : kMidPF;
  constexpr NarrowGroup<DT> GR{};
  constexpr int NSG = narrow_group_steps(DT);
  constexpr int NCB = mid_ncb(DT), NCBW = (NCB + NW - 1) / NW;
  constexpr int ZS = mid_zs(DT), RS = 16 * KB + 1, ROWS = 16 * NW, WG = 64 * NW;
  constexpr int ZI = (16 * DT + 63) / 64;
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int NBUF = MODE == 1 ? 1 : mid_nbuf(DT, KB);
  double* Zt = reinterpret_cast<double*>(smem);         // [NBUF][ROWS][ZS]
  double* Rt = Zt + (size_t)NBUF * ROWS * ZS;           // [NBUF][ROWS][RS]
  double* etab = Rt + (MODE == 1 ? 0 : (size_t)NBUF * ROWS * RS);         // [64]: 2^(i/64)  (label mode: no R tile)
  double* sred = etab + 64;                             // [NW]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, j = lane & 15;
  const int K = a.K;
  const int64_t N = a.N;

  for (int e = tid; e < NBUF * ROWS; e += WG) {         // the zero slots behind [z, 1]: written once
    Zt[(size_t)e * ZS + DT + 1] = 0.0; Zt[(size_t)e * ZS + DT + 2] = 0.0; Zt[(size_t)e * ZS + DT + 3] = 0.0;
  }
  if (tid < 64) etab[tid] = exp2((double)tid * (1.0 / 64.0));

  // rows of the wave's 16-row group: element e = lane + 64 i of the (16, DT) block
  auto zoff = [&](int i) -> int {                       // (recomputed where used: a division by a constant, not ZI live registers)
    const int e = lane + 64 * i, r = e / DT;
    return e < 16 * DT ? r * ZS + (e - r * DT) : -1;
  };
  const int64_t nss = (N + ROWS - 1) / ROWS;
  double zr[ZI];
  auto load_z = [&](int64_t T) {
    const int64_t base = (T * ROWS + 16 * wave) * DT, total = N * DT;
#pragma unroll
    for (int i = 0; i < ZI; ++i) {
      const int64_t gidx = base + lane + 64 * i;
      zr[i] = (lane + 64 * i < 16 * DT && gidx < total) ? a.Z[gidx] : 0.0;
    }
  };
  if ((int64_t)blockIdx.x < nss) load_z(blockIdx.x);

  // second product: byte offsets of the two factors of feature 16 cb + j (this wave's column blocks cb = wave + NW i) inside
  // row q of a group; a block past the last one reads the zero slot twice
  constexpr int NCBS = MODE == 1 ? 1 : NCBW;              // (label mode: no statistics in this kernel)
  int spa[NCBS], spb[NCBS];
#pragma unroll
  for (int i = 0; i < NCBS; ++i) {
    const int cb = wave + NW * i;
    const int f = 16 * cb + j;
    const bool real = cb < NCB;
    spa[i] = q * ZS + (real ? a.feat[2 * f] : DT + 1);
    spb[i] = q * ZS + (real ? a.feat[2 * f + 1] : DT + 1);
  }
  d4 sacc[KB][NCBS];
#pragma unroll
  for (int rb = 0; rb < KB; ++rb)
#pragma unroll
    for (int i = 0; i < NCBS; ++i) sacc[rb][i] = d4{0.0, 0.0, 0.0, 0.0};
  double sc_lse = 0.0, sc_prod = 1.0;
  int since_flush = 0;
  wg_sync();

  int buf = 0;
  for (int64_t T = blockIdx.x; T < nss; T += gridDim.x, buf ^= (NBUF - 1)) {
    double* Zw = Zt + ((size_t)buf * ROWS + 16 * wave) * ZS;
    const int64_t n = T * ROWS + 16 * wave + j;
    const bool valid = n < N;
    // ---- this wave's rows -> the shared z tile (the other buffer is still read by slower waves' second product)
#pragma unroll
    for (int i = 0; i < ZI; ++i) {
      const int zo = zoff(i);
      if (zo >= 0) Zw[zo] = zr[i];
    }
    if (q == 0) Zw[j * ZS + DT] = valid ? 1.0 : 0.0;    // rows past N: every feature 0 — nothing reaches the statistics
    if (T + gridDim.x < nss) load_z(T + gridDim.x);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---- L = Theta . Phi' over the grouped steps: the lane's row and its copy shifted by q, in registers
    d4 acc[KB];
#pragma unroll
    for (int rb = 0; rb < KB; ++rb) acc[rb] = d4{0.0, 0.0, 0.0, 0.0};
    {
      const double* zrow = Zw + j * ZS;
      const double* zsh = zrow + q;
      double zS[DT + 1];
#pragma unroll
      for (int i = 0; i <= DT; ++i) zS[i] = zsh[i];
      // opaque scalar base per super-step: slice addresses = scalar base + lane offset + immediates, not 2 NSG KB hoisted registers
      gptr_t thg = (gptr_t)a.theta;
      asm volatile("" : "+s"(thg));
      double ring[PF];
#pragma unroll
      for (int e = 0; e < PF; ++e) ring[e] = thg[e * 64 + lane];
#pragma unroll
      for (int s = 0; s < NSG; ++s) {
        const double bcur = zrow[GR.a[s]] * zS[GR.b0[s]];
#pragma unroll
        for (int rb = 0; rb < KB; ++rb) {
          const int e = s * KB + rb;
          const double av = ring[e % PF];
          ring[e % PF] = thg[(e + PF) * 64 + lane];                    // (the last reads take the zero slices behind the image)
          acc[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bcur, acc[rb], 0, 0, 0);
        }
        // (the matrix instructions form one dependent chain per row block: without a fence hipcc hoists every row read and
        //  every product of the 153 steps in front of it — 300 live registers, kilobytes of scratch)
        if (s % 4 == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- softmax over the row's components: this lane holds components 16 rb + 4 r + q of row j
    __builtin_amdgcn_s_setprio(2);
    double m;
    {
      double mv[4] = {acc[0][0], acc[0][1], acc[0][2], acc[0][3]};
#pragma unroll
      for (int rb = 1; rb < KB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) mv[r] = fmax(mv[r], acc[rb][r]);
      m = fmax(fmax(mv[0], mv[1]), fmax(mv[2], mv[3]));
      m = fmax(m, __shfl_xor(m, 16));
      m = fmax(m, __shfl_xor(m, 32));
    }
    if constexpr (MODE == 1) {
      // inclusive cumulative sums of e = exp(l - max) over the lane's quarter (component q V + 4 rb + r sits in acc[rb][r]), then over
      // the four quarters of the row (lanes j, j + 16, j + 32, j + 48); label = #{k : u cum_K > cum_k}
      double run = 0.0;
#pragma unroll
      for (int rb = 0; rb < KB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          run += exp_nonpos(acc[rb][r] - m, etab);
          acc[rb][r] = run;
        }
      double incl = run;
      {
        double v = __shfl_up(incl, 16);  if (q >= 1) incl += v;
        v = __shfl_up(incl, 32);         if (q >= 2) incl += v;
      }
      double excl = __shfl_up(incl, 16);
      if (q == 0) excl = 0.0;
      const double ctot = __shfl(incl, 48 + j);
      const double uu = a.u ? (valid ? a.u[n] : 0.0) : philox_uniform(a.seed, (uint64_t)(a.row0 + n), a.sweep);
      const double tl = uu * ctot - excl;
      int cnt = 0;
#pragma unroll
      for (int rb = 0; rb < KB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) cnt += tl > acc[rb][r] ? 1 : 0;
      cnt += __shfl_xor(cnt, 16);
      cnt += __shfl_xor(cnt, 32);
      if (q == 0 && valid) a.labels[n] = cnt < K ? cnt : K - 1;
      __builtin_amdgcn_s_setprio(0);
      continue;
    }
    double sv[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int rb = 0; rb < KB; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[rb][r] = exp_nonpos(acc[rb][r] - m, etab);
        sv[r] += acc[rb][r];
      }
    double ssum = (sv[0] + sv[1]) + (sv[2] + sv[3]);
    ssum += __shfl_xor(ssum, 16);
    ssum += __shfl_xor(ssum, 32);
    double inv = __builtin_amdgcn_rcp(ssum);
    inv = fma(fma(-ssum, inv, 1.0), inv, inv);
    inv = fma(fma(-ssum, inv, 1.0), inv, inv);
    if (q == 0 && valid) { sc_lse += m; sc_prod *= ssum; }
    // per-row weights of the statistics (mimo_estep_weighted; the NaN-row mask): tables and scalars stay unweighted (hgmm.py:199-207)
    if (a.u) inv *= valid ? a.u[n] : 0.0;
    {
      double* rw = Rt + ((size_t)buf * ROWS + 16 * wave + j) * RS + q;
#pragma unroll
      for (int rb = 0; rb < KB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) rw[16 * rb + 4 * r] = acc[rb][r] * inv;
    }
    __builtin_amdgcn_s_setprio(0);
    wg_sync();                                          // every wave's rows and responsibilities of this super-step are in LDS

    // ---- S += R . Phi over all 16 NW rows, this wave's column blocks
    const double* Zb = Zt + (size_t)buf * ROWS * ZS;
    const double* Rb = Rt + (size_t)buf * ROWS * RS + (size_t)q * RS + j;
#pragma unroll
    for (int g = 0; g < NW; ++g) {
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        const double* zt = Zb + (size_t)(16 * g + 4 * t4) * ZS;
        double bv[NCBW], av[KB];
#pragma unroll
        for (int i = 0; i < NCBW; ++i) bv[i] = zt[spa[i]] * zt[spb[i]];
#pragma unroll
        for (int rb = 0; rb < KB; ++rb) av[rb] = Rb[(size_t)(16 * g + 4 * t4) * RS + 16 * rb];
#pragma unroll
        for (int rb = 0; rb < KB; ++rb)
#pragma unroll
          for (int i = 0; i < NCBW; ++i)
            sacc[rb][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[rb], bv[i], sacc[rb][i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);              // (as above: one group of operands in flight, not all 4 NW of them)
      }
    }
    if constexpr (NBUF == 1) wg_sync();                 // single tiles: the next super-step's rows overwrite what the slowest wave still reads
    if (++since_flush == 32) {           // 128^32 = 2^224 stays inside the float64 range
      sc_lse += log(sc_prod);
      sc_prod = 1.0;
      since_flush = 0;
    }
  }

  if constexpr (MODE == 1) return;
  // ---- per-workgroup partial block: every column block has exactly one owner
  const int FT = a.F16_total;
  const size_t pstride = (size_t)a.K16 * 16 * FT + 4;
  double* P = a.partials + (size_t)blockIdx.x * pstride;
#pragma unroll
  for (int i = 0; i < NCBW; ++i) {
    const int cb = wave + NW * i;
    if (cb < NCB) {
#pragma unroll
      for (int rb = 0; rb < KB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = 16 * rb + 4 * r + q;
          P[(size_t)k * FT + 16 * cb + j] = k < K ? sacc[rb][i][r] : 0.0;
        }
    }
  }
  sc_lse += log(sc_prod);
  sc_lse = wave_sum(sc_lse);
  wg_sync();
  if (lane == 0) sred[wave] = sc_lse;
  wg_sync();
  if (tid == 0 && a.write_scalars) {
    double* Ps = P + (size_t)a.K16 * 16 * FT;
    double s2 = 0.0;
    for (int w = 0; w < NW; ++w) s2 += sred[w];
    Ps[0] = s2; Ps[1] = 0.0; Ps[2] = 0.0; Ps[3] = 0.0;
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
typedef void (*mid_fn)(const KernelArgs);
template <int D>
static mid_fn pick_mid_d(int kb) {
  if constexpr (mid_exists(D, 1)) { if (kb == 1) return mid_kernel<D, 1, mid_waves(D, 1)>; }
  if constexpr (mid_exists(D, 2)) { if (kb == 2) return mid_kernel<D, 2, mid_waves(D, 2)>; }
  if constexpr (mid_exists(D, 3)) { if (kb == 3) return mid_kernel<D, 3, mid_waves(D, 3)>; }
  if constexpr (mid_exists(D, 4)) { if (kb == 4) return mid_kernel<D, 4, mid_waves(D, 4)>; }
  if constexpr (mid_exists(D, 5)) { if (kb == 5) return mid_kernel<D, 5, mid_waves(D, 5)>; }
  if constexpr (mid_exists(D, 6)) { if (kb == 6) return mid_kernel<D, 6, mid_waves(D, 6)>; }
  if constexpr (mid_exists(D, 7)) { if (kb == 7) return mid_kernel<D, 7, mid_waves(D, 7)>; }
  if constexpr (mid_exists(D, 8)) { if (kb == 8) return mid_kernel<D, 8, mid_waves(D, 8)>; }
  return nullptr;
}
static mid_fn pick_mid(int D, int kb) {
  switch (D) {
#define MIMO_MD(d) case d: return pick_mid_d<d>(kb);
    MIMO_MD(5) MIMO_MD(6) MIMO_MD(7) MIMO_MD(8) MIMO_MD(9) MIMO_MD(10) MIMO_MD(11) MIMO_MD(12) MIMO_MD(13) MIMO_MD(14) MIMO_MD(15) MIMO_MD(16)
    MIMO_MD(17) MIMO_MD(18) MIMO_MD(19) MIMO_MD(20) MIMO_MD(21) MIMO_MD(22) MIMO_MD(23) MIMO_MD(24) MIMO_MD(25) MIMO_MD(26) MIMO_MD(27)
    MIMO_MD(28) MIMO_MD(29) MIMO_MD(30) MIMO_MD(31) MIMO_MD(32)
#undef MIMO_MD
  }
  return nullptr;
}

// label-draw instantiations: K <= 48 (one to three row blocks), four waves
static mid_fn pick_mid_labels(int D, int kb) {
  switch (D) {
#define MIMO_ML(d) case d: return kb == 1 ? mid_kernel<d, 1, 4, 1> : kb == 2 ? mid_kernel<d, 2, 4, 1> : kb == 3 ? mid_kernel<d, 3, 4, 1> : nullptr;
    MIMO_ML(10) MIMO_ML(11) MIMO_ML(12) MIMO_ML(13) MIMO_ML(14) MIMO_ML(15) MIMO_ML(16)
    MIMO_ML(17) MIMO_ML(18) MIMO_ML(19) MIMO_ML(20) MIMO_ML(21) MIMO_ML(22) MIMO_ML(23) MIMO_ML(24) MIMO_ML(25) MIMO_ML(26) MIMO_ML(27)
    MIMO_ML(28) MIMO_ML(29) MIMO_ML(30) MIMO_ML(31) MIMO_ML(32)
#undef MIMO_ML
  }
  return nullptr;
}
bool mid_labels_covers(int K, int D, int structure) {
  static const bool on = [] { const char* e = getenv("MIMO_MID_LABELS"); return !e || atoi(e) != 0; }();       // tuning knob
  return on && structure == 0 && K >= 1 && K <= 48 && pick_mid_labels(D, (K + 15) / 16) != nullptr;
}
int mid_labels_grid(const KernelArgs& a, int num_cu) {
  const int64_t need = (a.N + 63) / 64;
  int64_t g = (int64_t)num_cu * 2;
  if (g > need) g = need;
  return (int)(g < 1 ? 1 : g);
}
hipError_t launch_mid_labels(const KernelArgs& a, int grid, hipStream_t stream) {
  const int kb = (a.K + 15) / 16;
  mid_fn fn = pick_mid_labels(a.D, kb);
  if (!fn || a.K16 != kb || !a.labels) return hipErrorInvalidValue;
  const size_t lds = sizeof(double) * ((size_t)64 * mid_zs(a.D) + 64 + 8);
  hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, stream, a);
  return hipGetLastError();
}

// Which (K, Dz) the kernel exists for: full feature map, K <= 32, Dz = 9 .. 32 (MIMO_MID=0: route off; the router in
// mimo_abi.cpp decides where it is preferred over the narrow / tile / row-owner kernels)
bool mid_covers(int K, int D, int structure) {
  static const bool on = [] { const char* e = getenv("MIMO_MID"); return !e || atoi(e) != 0; }();       // tuning knob
  return on && structure == 0 && K >= 1 && K <= 128 && D >= 5 && D <= 32 && pick_mid(D, (K + 15) / 16) != nullptr;
}
int mid_steps(int D) { return narrow_group_steps(D); }
int mid_pf() { return 16; }            // zero slices behind the image: the deepest ring
static int mid_waves_rt(int D, int kb);
int mid_rows_per_step(int K, int D) { return 16 * mid_waves_rt(D, (K + 15) / 16); }

template <int D>
static void mid_geom_d(int kb, int* waves, int* wgs, size_t* lds) {
  switch (kb) {
#define MIMO_MG(k) case k: *waves = mid_waves(D, k); *wgs = mid_wgs_per_cu(D, k); *lds = mid_lds_bytes(D, k); return;
    MIMO_MG(1) MIMO_MG(2) MIMO_MG(3) MIMO_MG(4) MIMO_MG(5) MIMO_MG(6) MIMO_MG(7) MIMO_MG(8)
#undef MIMO_MG
  }
  *waves = 4; *wgs = 1; *lds = 0;
}
static void mid_geom(int D, int kb, int* waves, int* wgs, size_t* lds) {
  switch (D) {
#define MIMO_MD(d) case d: mid_geom_d<d>(kb, waves, wgs, lds); return;
    MIMO_MD(5) MIMO_MD(6) MIMO_MD(7) MIMO_MD(8) MIMO_MD(9) MIMO_MD(10) MIMO_MD(11) MIMO_MD(12) MIMO_MD(13) MIMO_MD(14) MIMO_MD(15) MIMO_MD(16)
    MIMO_MD(17) MIMO_MD(18) MIMO_MD(19) MIMO_MD(20) MIMO_MD(21) MIMO_MD(22) MIMO_MD(23) MIMO_MD(24) MIMO_MD(25) MIMO_MD(26) MIMO_MD(27)
    MIMO_MD(28) MIMO_MD(29) MIMO_MD(30) MIMO_MD(31) MIMO_MD(32)
#undef MIMO_MD
  }
  *waves = 4; *wgs = 1; *lds = 0;
}
static int mid_waves_rt(int D, int kb) { int w, g; size_t l; mid_geom(D, kb, &w, &g, &l); return w; }

int mid_grid(const KernelArgs& a, int num_cu) {
  const int kb = (a.K + 15) / 16;
  int nw, per_cu; size_t ldsb;
  mid_geom(a.D, kb, &nw, &per_cu, &ldsb);
  const int64_t need = (a.N + 16 * nw - 1) / (16 * nw);
  int64_t g = (int64_t)num_cu * per_cu;
  if (g > need) g = need;
  return (int)(g < 1 ? 1 : g);
}

hipError_t launch_mid(const KernelArgs& a, int grid, hipStream_t stream) {
  const int kb = (a.K + 15) / 16;
  mid_fn fn = pick_mid(a.D, kb);
  if (!fn || a.K > 128 || a.K16 != kb) return hipErrorInvalidValue;
  int nw, per_cu; size_t lds;
  mid_geom(a.D, kb, &nw, &per_cu, &lds);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * nw), lds, stream, a);
  return hipGetLastError();
}

}  // namespace mimo
