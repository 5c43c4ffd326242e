// The two-stage shapes (Dz > 16, or K > 64 at Dz >= 10): E-step and statistics kernels without latency-bound phases.
//
// wide_stats_kernel: statistics S = R . Phi of a K-major weight table.  Why a second kernel next to
// fused_kernel<.., kModeWeights>: at Dz = 32, K = 128 that one runs as two 4-wave
// workgroups per CU, each staging a 32 x 128 weight tile through LDS and building 6 column blocks of features per
// tile, 6 launches per sweep.  The per-phase trace (tools/stamps_chunked.py) shows the two workgroups of a CU in lock
// step: both wait out the weight tile's HBM round trip and the feature build's LDS chains together (5.1k cycles,
// matrix pipe idle), then share the pipe for their 2 x 96 MFMAs (12.3k) — 66 % of the FP64 rate.  Priorities and a
// start-up stagger do not separate them (a build that overlaps the neighbour's matrix phase gets one VALU issue slot
// per 64-cycle MFMA and falls back into step), so the idle phase itself has to go:
//   * ONE 8-wave workgroup per CU; wave w owns row block w % RBN and every CP-th column block (CP = 8 / RBN) of the
//     launch: up to 12 accumulator blocks per wave, 12 (RBN = 8) or 16 (RBN = 4) column blocks per launch — the table is
//     read 3 x (K = 128, Dz = 32) instead of 6 x, Z four times instead of seven; 128 < K <= 256: two row blocks per wave;
//   * the weights never touch LDS: lane (j, q) of the wave that owns row block rb loads the 8 consecutive rows
//     8q .. 8q+7 of component 16 rb + j — 64 contiguous bytes — and register s IS the MFMA A operand of contraction
//     step s (rows s, s+8, s+16, s+24); the next tile's 8 registers are in flight during this tile's matrix phase;
//   * z rows travel ahead in registers (double-buffered z tile) and the NEXT tile's features are built between this
//     tile's matrix steps into a second feature tile: one barrier per tile, no global or LDS round trip in front of an
//     MFMA; the tile loop is branch-free (clamped addresses instead of bounds tests).
// wide_estep_kernel (below): the softmax / label-draw pass that writes the table or the labels.
// Partial blocks, feature table and reduction are those of the tile kernels (mimo_kernels.hip).
#include "mimo_device.h"
#include "mimo_extra.h"

#include <type_traits>

namespace mimo {

#ifdef MIMO_STAMPS
// diagnostic build: per-wave cycle sums of the phases (tools/stamps_chunked.py)
#define WSTAMP(i)                                                                         \
  do {                                                                                    \
    unsigned long long t_;                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    st_sum[i] += t_ - st_prev;                                                            \
    st_prev = t_;                                                                         \
  } while (0)
#else
#define WSTAMP(i) do {} while (0)
#endif

constexpr int kWideWG = 512;       // 8 wavefronts
constexpr int kWideNCBL = 12;      // accumulator column blocks per wave, at most

typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));   // 16-byte load of an 8-byte aligned address

template <int RBN, int NCBL, int RBW = 1>
__global__ __launch_bounds__(kWideWG) void wide_stats_kernel(const KernelArgs a) {
  // RBW row blocks per wave (rb, rb + RBN, ...): K <= 128 runs RBW = 1; K <= 256 RBW = 2 with RBN = 8 — every B operand
  // then feeds two MFMAs, at twice the weight registers
  constexpr int T = kTile, CP = 8 / RBN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int D = a.D, K = a.K, K16 = a.K16;
  const int ncb = a.F16 / 16;                       // column blocks of this launch: NCBL * CP, or a few less (the
                                                    // surplus blocks are built from offset 0 and never stored)
  constexpr int ZS = 35;                            // fixed (odd, >= kMaxD + 2): row offsets of the feature build are immediates
  constexpr int RS = 16 * NCBL * CP + 2;            // fixed (every operand offset is an immediate); = 2 mod 4: the four row groups of a B-operand read fall into disjoint banks
  const int64_t N = a.N;

  extern __shared__ __align__(16) unsigned char smem[];
  double* Zs = reinterpret_cast<double*>(smem);               // [2][T][ZS]
  double* Ph = Zs + 2 * T * ZS;                                // [2][T][RS]  feature tiles of this and the next tile
  uint32_t* fo = reinterpret_cast<uint32_t*>(Ph + 2 * T * RS);     // [16 ncb]  byte offsets (a | b << 16) into a z~ row

  const uint8_t* featp = a.feat + 32 * a.cb0;
  for (int e = tid; e < 16 * ncb; e += kWideWG) fo[e] = 8u * featp[2 * e] | (8u * featp[2 * e + 1]) << 16;

  // The tile loop is branch-free straight-line code: every load is unconditional from a clamped (valid) address and
  // masked by a select where it matters.  (With the bounds tests as branches hipcc put `s_waitcnt vmcnt(0)` at the
  // joins — right behind the prefetch of the NEXT tile's weights, i.e. one HBM round trip per tile.)
  //   * the loop runs over the FULL tiles only; prefetches past the last one are clamped to it (what they fetch is
  //     never used); the partial tile at the end, if any, is one workgroup's plain epilogue (tail_tile);
  //   * padding components (k >= K) read component K - 1's weights; their rows of the partial block are written as 0;
  //   * waves whose row block is past K16 run the same matrix steps on clamped operands and store nothing.
  const int64_t nfull = N / T, last_tile = nfull - 1, total = N * D;
  // z staging: T*D <= 1024 elements, 2 per thread; registers hold the tile three iterations ahead
  int zoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + kWideWG * i, pt = e / D;
    zoff[i] = e < T * D ? pt * ZS + (e - pt * D) : -1;
  }
  double zr[2];
  auto load_z = [&](int64_t t) {
    const int64_t base = (t < last_tile ? t : last_tile) * T * D;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t g = base + tid + kWideWG * i;
      zr[i] = a.Z[g < total ? g : total - 1];        // (threads without an element: zoff < 0, value unused)
    }
  };
  auto store_z = [&](double* Zb) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (zoff[i] >= 0) Zb[zoff[i]] = zr[i];
    if (tid < T) {
      Zb[tid * ZS + D] = 1.0;
      Zb[tid * ZS + D + 1] = 0.0;                            // padded features read this slot
    }
  };

  // matrix role of this wave
  const int rb = wave % RBN, cpart = wave / RBN;
  const double* wrow[RBW];
#pragma unroll
  for (int r = 0; r < RBW; ++r) {
    const int kcomp = 16 * (rb + RBN * r) + j;
    wrow[r] = a.resp + (int64_t)(kcomp < K ? kcomp : K - 1) * N;
  }
  // A operands of one tile: rows 8q .. 8q+7 of this lane's component(s), 64 contiguous bytes each
  auto load_w = [&](int64_t t, double (&w)[RBW][8]) {
    const int64_t off = (t < last_tile ? t : last_tile) * T + 8 * q;
#pragma unroll
    for (int r = 0; r < RBW; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const d2u v = *reinterpret_cast<const d2u*>(wrow[r] + off + 2 * i);
        w[r][2 * i] = v[0]; w[r][2 * i + 1] = v[1];
      }
  };

  d4 acc[RBW][NCBL];
#pragma unroll
  for (int r = 0; r < RBW; ++r)
#pragma unroll
    for (int i = 0; i < NCBL; ++i) acc[r][i] = d4{0.0, 0.0, 0.0, 0.0};

  // feature build: thread (fcol, rq) makes the feature columns fcol, fcol + NCOLS, ... for the NR rows NR rq .. of the tile
  // (NR = 4 with 64-column groups where the column count allows, else 2 with 32-column groups): one (a, b) offset pair
  // per NR products, the rows at immediate offsets.  Group g is read after matrix step g and multiplied / stored after
  // step g + 1 of the PREVIOUS tile: the build's LDS round trips pass under NCBL MFMAs each instead of holding all
  // eight waves at a barrier, and one barrier per tile is left.
  constexpr int NBF = NCBL * CP;                     // products per thread
  constexpr int NR = NBF % 4 == 0 ? 4 : 2, NCOLS = kWideWG * NR / T, NG = 16 * NBF / NCOLS;
  static_assert(NG >= 1 && NG <= 7 && NG * NCOLS == 16 * NBF, "group g: read after step g, stored after step g + 1");
  const int fcol = tid % NCOLS, rq = tid / NCOLS;
  double* Ph1 = Ph + T * RS;                         // second feature tile

  double wa[RBW][8], wb[RBW][8];
  // this thread's (a, b) byte offsets stay in registers: a table read inside the matrix phase would put an LDS round
  // trip (s_waitcnt in the in-order instruction stream) in front of the following MFMAs
  uint32_t w2[NG];
  wg_sync();
#pragma unroll
  for (int g = 0; g < NG; ++g) w2[g] = NCOLS * g + fcol < 16 * ncb ? fo[NCOLS * g + fcol] : 0u;   // columns past ncb: offsets 0, a finite product nobody stores
  double za[2][NR], zb[2][NR];
  auto build_loads = [&](int g, const double* zb_) {
    const unsigned char* zq = reinterpret_cast<const unsigned char*>(zb_ + NR * rq * ZS);
    const unsigned char* pa = zq + (w2[g] & 0xFFFFu);
    const unsigned char* pb = zq + (w2[g] >> 16);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      za[g & 1][r] = *reinterpret_cast<const double*>(pa + r * ZS * 8);
      zb[g & 1][r] = *reinterpret_cast<const double*>(pb + r * ZS * 8);
    }
  };
  auto build_stores = [&](int g, double* pb_) {
    double* prow = pb_ + NR * rq * RS + NCOLS * g + fcol;
#pragma unroll
    for (int r = 0; r < NR; ++r) prow[r * RS] = za[g & 1][r] * zb[g & 1][r];
  };

  const int64_t G = gridDim.x;
  int cur = 0;
  // iteration i (tile t): matrix steps on Phi(t) in Ph[cur]; Phi(t + G) built from Zs[cur ^ 1] into Ph[cur ^ 1];
  // z~(t + 2G) staged into Zs[cur]; z(t + 3G) fetched
  auto tile = [&](int64_t t, double (&wc)[RBW][8], double (&wn)[RBW][8]) {
    const double* Pc = cur ? Ph1 : Ph;
    double* Pn = cur ? Ph : Ph1;
    const double* Zn = Zs + (cur ^ 1) * T * ZS;
    load_w(t + G, wn);
    int ph_off = 8 * q * RS + 16 * cpart + j;
    asm volatile("" : "+v"(ph_off));
    const double* phq = Pc + ph_off;
    double bv[NCBL];
#pragma unroll
    for (int i = 0; i < NCBL; ++i) bv[i] = phq[16 * CP * i];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      // S += R . Phi: step s contracts rows {s, s+8, s+16, s+24}; A = this wave's weight registers,
      // B lane (kk = q, col j) = Phi[8q + s][16 cb + j]
#pragma unroll
      for (int i = 0; i < NCBL; ++i) {
#pragma unroll
        for (int r = 0; r < RBW; ++r) acc[r][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(wc[r][s], bv[i], acc[r][i], 0, 0, 0);
        // operand of the next step: NCBL MFMAs ahead of its use.  (What-if: with every second operand a copy of its
        // neighbour — half the LDS reads, wrong results — a launch at C5's shape took 1.495 instead of 1.522 ms: LDS
        // operand traffic is not what bounds this kernel.)
        if (s + 1 < 8) bv[i] = phq[(s + 1) * RS + 16 * CP * i];
      }
      __builtin_amdgcn_sched_barrier(0);      // no hoisting of later steps' reads: 8 x NCBL operands do not fit
      if (s >= 1 && s - 1 < NG) build_stores(s - 1, Pn);
      if (s < NG) build_loads(s, Zn);
      __builtin_amdgcn_sched_barrier(0);
    }
    store_z(Zs + cur * T * ZS);               // z~(t + 2G), fetched one iteration ago
    load_z(t + 3 * G);
    cur ^= 1;
    wg_sync();
  };
  if (blockIdx.x < nfull) {
    // prologue: z~ of the first two tiles, Phi of the first
    load_z(blockIdx.x);
    store_z(Zs);
    load_z(blockIdx.x + G);
    store_z(Zs + T * ZS);
    load_z(blockIdx.x + 2 * G);
    load_w(blockIdx.x, wa);
    wg_sync();
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      build_loads(g, Zs);
      build_stores(g, Ph);
    }
    wg_sync();
    int64_t t = blockIdx.x;
    for (; t + G < nfull; t += 2 * G) {
      tile(t, wa, wb);
      tile(t + G, wb, wa);
    }
    if (t < nfull) tile(t, wa, wb);
  }
  // the partial tile behind the last full one: plain, unpipelined
  if (a.ntiles > nfull && (int64_t)blockIdx.x == nfull % G) {
    const int64_t n0 = nfull * T;
    for (int e = tid; e < T * D; e += kWideWG) {
      const int pt = e / D;
      Zs[pt * ZS + (e - pt * D)] = n0 * D + e < total ? a.Z[n0 * D + e] : 0.0;
    }
    if (tid < T) {
      Zs[tid * ZS + D] = n0 + tid < N ? 1.0 : 0.0;           // rows past N contribute nothing
      Zs[tid * ZS + D + 1] = 0.0;
    }
    wg_sync();
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      build_loads(g, Zs);
      build_stores(g, Ph);
    }
    wg_sync();
    const double* phq = Ph + 8 * q * RS + 16 * cpart + j;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int64_t n = n0 + 8 * q + s;
#pragma unroll
      for (int r = 0; r < RBW; ++r) {
        const double w = n < N ? wrow[r][n] : 0.0;
#pragma unroll
        for (int i = 0; i < NCBL; ++i)
          acc[r][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(w, phq[s * RS + 16 * CP * i], acc[r][i], 0, 0, 0);
      }
    }
  }

  // ---- per-workgroup partial block (layout of the tile kernels)
  const int FT = a.F16_total, Kpad = 16 * K16;
  const size_t pstride = (size_t)Kpad * FT + 4;
  double* P = a.partials + (size_t)blockIdx.x * pstride + 16 * a.cb0;
#pragma unroll
  for (int rr = 0; rr < RBW; ++rr) {
    const int rbi = rb + RBN * rr;
    if (rbi < K16) {
#pragma unroll
      for (int i = 0; i < NCBL; ++i) {
        const int cb = cpart + CP * i;
        if (cb < ncb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int k = 16 * rbi + q + 4 * r;
            P[(size_t)k * FT + 16 * cb + j] = k < K ? acc[rr][i][r] : 0.0;
          }
        }
      }
    }
  }
  if (tid == 0 && a.write_scalars) {
    double* Ps = a.partials + (size_t)blockIdx.x * pstride + (size_t)Kpad * FT;
    Ps[0] = 0.0; Ps[1] = 0.0; Ps[2] = 0.0; Ps[3] = 0.0;
  }
}

// ------------------------------------------------------------------------------------------
// E-step of the wide shapes (softmax pass, no statistics): estep_chunked_kernel's blocking — two 4-wave workgroups
// per CU, wave w owns row blocks w, w + 4, ... (RBW of them) and both 16-row column groups of a 32-row tile — without
// its latency-bound phases:
//   * the Theta slices stream from the L2-resident image through a 6-deep register ring (what-if build: serving every
//     slice from one cache line changes nothing, so L2 is not the bound);
//   * the feature tile is built chunk by chunk (96 features = 24 contraction steps) into a double buffer: table
//     offsets after matrix step 2, operand reads after steps 6 / 14, products and stores after steps 10 / 18 of the PREVIOUS
//     chunk — one barrier per chunk, no LDS round trip in front of an MFMA, z rows two tiles ahead in registers;
//     what is left of a chunk barrier's bubble is filled by the other workgroup of the CU (an 8-wave variant of
//     this kernel, one workgroup per CU, stalled all eight waves there: 16 % of the tile time in barrier waits);
//   * the softmax runs on the accumulators (C layout: lane (q, j), register r = component q + 4r of the wave's row
//     block, datum j of the column group): per-datum max / sum / sum e l of the wave's 16 RBW components by two
//     cross-row exchanges, the four waves' partials through 3 x 4 x 32 doubles of LDS in a fixed order, the table
//     written from the registers (16 consecutive rows = one 128-byte line per component and column group).
// Softmax table or label draw; scalars and tables as estep_chunked_kernel's.
// ------------------------------------------------------------------------------------------
constexpr int kWideEstepCF = 96;       // features per chunk
constexpr int kWideEstepZS = 35;       // z~ row stride (doubles): odd, >= kMaxD + 2
template <int RBW>
__global__ __launch_bounds__(kWG, 2) void wide_estep_kernel(const KernelArgs a) {
  constexpr int T = kTile, CF = kWideEstepCF, NSc = CF / 4, RSc = CF + 2, PF = 6, NBF = CF * T / kWG, NW = kWG / 64, ZPT = 4;
  static_assert(NBF * kWG == CF * T && (NSc * RBW) % PF == 0, "chunk geometry");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int D = a.D, K = a.K, K16 = a.K16;
  constexpr int ZS = kWideEstepZS;                             // fixed (odd, >= Dz + 2): row offsets of the feature build are immediates
  const int nchunk = (a.F16 + CF - 1) / CF, NSP = chunked_ns_pad(a.F16);     // row-block stride of the Theta image
  const int64_t N = a.N, total = N * D, G = gridDim.x;

  extern __shared__ __align__(16) unsigned char smem[];
  double* Zs = reinterpret_cast<double*>(smem);               // [2][T][ZS]
  double* Ph = Zs + 2 * T * ZS;                                // [2][T][RSc]  feature chunks
  double* red = Ph + 2 * T * RSc;                              // [3][NW][T]  per-wave partial max / sum / sum e l
  double* rbs = red + 3 * NW * T;                              // [16][T]     label draw: sum of e per row block
  double* ured = rbs + 16 * T;                                 // [T]         ... uniforms of the tile's rows
  int* cred = reinterpret_cast<int*>(ured + T);                // [NW][T]     ... per-wave counts
  double* etab = reinterpret_cast<double*>(cred + NW * T);                              // [64]
  uint32_t* fo = reinterpret_cast<uint32_t*>(etab + 64);       // [nchunk CF]  byte offsets (a | b << 16) into a z~ row
  if (tid < 64) etab[tid] = exp2((double)tid * (1.0 / 64.0));
  for (int e = tid; e < nchunk * CF; e += kWG) {
    const uint32_t fa = e < a.F16 ? a.feat[2 * e] : (uint32_t)(D + 1), fb = e < a.F16 ? a.feat[2 * e + 1] : (uint32_t)(D + 1);
    fo[e] = 8u * fa | (8u * fb) << 16;
  }

  // z staging (as wide_stats_kernel): registers hold the tile after next; rows past N are zero with a zero "1" slot
  int zoff[ZPT];
#pragma unroll
  for (int i = 0; i < ZPT; ++i) {
    const int e = tid + kWG * i, pt = e / D;
    zoff[i] = e < T * D ? pt * ZS + (e - pt * D) : -1;
  }
  double zr[ZPT];
  auto load_z = [&](int64_t t) {
    const int64_t base = t * T * D;
#pragma unroll
    for (int i = 0; i < ZPT; ++i) {
      const int64_t g = base + tid + kWG * i;
      zr[i] = a.Z[g < total ? g : total - 1];
    }
  };
  auto store_z = [&](int64_t t, double* Zb) {
    const int64_t base = t * T * D;
#pragma unroll
    for (int i = 0; i < ZPT; ++i)
      if (zoff[i] >= 0) Zb[zoff[i]] = base + tid + kWG * i < total ? zr[i] : 0.0;
    if (tid < T) {
      Zb[tid * ZS + D] = t * T + tid < N ? 1.0 : 0.0;
      Zb[tid * ZS + D + 1] = 0.0;
    }
  };

  // feature build of one chunk: thread (fcol = tid & 31, rq = tid >> 5) makes the features fcol, fcol + 32, fcol + 64 of the
  // chunk for the four rows 4 rq .. 4 rq + 3: one (a, b) offset pair per feature — two address adds — and the rows at
  // immediate offsets (ZS and RSc are compile-time), 1.5 VALU instructions per product instead of 3 with a thread per row
  const int fcol = tid & 31, rq = tid >> 5;
  constexpr int NFG = CF / 32;                                 // feature groups of 32 columns per chunk
  static_assert(NFG == 3 && T == 32 && kWG == 256, "build mapping");
  uint32_t w2[NFG];
  double za[4], zb[4];
  auto build_offsets = [&](int ch) {
#pragma unroll
    for (int m = 0; m < NFG; ++m) w2[m] = fo[ch * CF + 32 * m + fcol];
  };
  auto build_loads = [&](int m, const double* zb_) {
    const unsigned char* zq = reinterpret_cast<const unsigned char*>(zb_ + 4 * rq * ZS);
    const unsigned char* pa = zq + (w2[m] & 0xFFFFu);
    const unsigned char* pb = zq + (w2[m] >> 16);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      za[r] = *reinterpret_cast<const double*>(pa + r * ZS * 8);
      zb[r] = *reinterpret_cast<const double*>(pb + r * ZS * 8);
    }
  };
  auto build_stores = [&](int m, double* pb_) {
    double* prow = pb_ + 4 * rq * RSc + 32 * m + fcol;
#pragma unroll
    for (int r = 0; r < 4; ++r) prow[r * RSc] = za[r] * zb[r];
  };

  // Theta stream of this wave: element e = s * RBW + i of a tile is slice s of row block wave + 4 i (a row block past
  // K16 streams the last one's slices: its accumulators are overwritten with the padding value before the softmax)
  gptr_t thb[RBW];
#pragma unroll
  for (int i = 0; i < RBW; ++i) {
    const int rbi = wave + NW * i < K16 ? wave + NW * i : K16 - 1;
    thb[i] = (gptr_t)(a.theta + (size_t)rbi * NSP * 64);       // scalar base: the loads use base + lane offset + immediate
  }
  auto slice = [&](int ch, int ee) {          // element ee (0 .. NSc RBW - 1) of chunk ch
    return thb[ee % RBW][((size_t)ch * NSc + ee / RBW) * 64 + lane];
  };
  double ring[PF];
  double sc_lse = 0.0, sc_rl = 0.0;
  const bool want_sel = a.split != 0;        // sum_k r l only feeds the entropy split of the ELBO scalars
  const bool single = nchunk == 1;          // F16 <= 96 (Dz = 10 .. 12)
  const bool gibbs = a.gibbs != 0;

  if ((int64_t)blockIdx.x < a.ntiles) {
    load_z(blockIdx.x);
    store_z(blockIdx.x, Zs);
    load_z(blockIdx.x + G);
    if (single) {                             // one chunk per tile: the rows are staged two tiles ahead (see the tile loop)
      store_z(blockIdx.x + G, Zs + T * ZS);
      load_z(blockIdx.x + 2 * G);
    }
#pragma unroll
    for (int e = 0; e < PF; ++e) ring[e] = slice(0, e);
    wg_sync();
    build_offsets(0);
#pragma unroll
    for (int m = 0; m < NFG; ++m) {
      build_loads(m, Zs);
      build_stores(m, Ph);
    }
    wg_sync();
  }
  int cur = 0, pbuf = 0;
#ifdef MIMO_STAMPS
  unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
  for (int64_t t = blockIdx.x; t < a.ntiles; t += G) {
    const int64_t n0 = t * T;
    d4 acc[RBW][2];
#pragma unroll
    for (int i = 0; i < RBW; ++i) { acc[i][0] = d4{0.0, 0.0, 0.0, 0.0}; acc[i][1] = d4{0.0, 0.0, 0.0, 0.0}; }
    for (int ch = 0; ch < nchunk; ++ch) {
      const bool last = ch + 1 == nchunk;
      const int nch = last ? 0 : ch + 1;                                   // chunk built during this one
      const double* Zn = Zs + (last ? cur ^ 1 : cur) * T * ZS;             // ... from this tile's rows or the next tile's
      const double* Pc = Ph + pbuf * T * RSc;
      double* Pn = Ph + (pbuf ^ 1) * T * RSc;
      // scalar bases of this chunk's Theta slices and of the first of the next chunk's, one per window of 8 slices
      // (forced into SGPRs: a load is then base + lane offset + an immediate below 4 KB, not a 64-bit VALU address)
      constexpr int NWIN = (NSc + 7) / 8;
      gptr_t cbase[RBW][NWIN], nbase[RBW];
#pragma unroll
      for (int i = 0; i < RBW; ++i) {
#pragma unroll
        for (int w = 0; w < NWIN; ++w) {
          cbase[i][w] = thb[i] + ((size_t)ch * NSc + 8 * w) * 64;
          asm volatile("" : "+s"(cbase[i][w]));
        }
        nbase[i] = thb[i] + (size_t)nch * NSc * 64;
        asm volatile("" : "+s"(nbase[i]));
      }
      int p_off = j * RSc + q;
      asm volatile("" : "+v"(p_off));
      const double* p0 = Pc + p_off;
      const double* p1 = p0 + 16 * RSc;
      // B operands are read three contraction steps ahead of their MFMAs; the sched_barriers keep hipcc from sinking
      // the reads back to their use (an LDS read is 2-3 MFMA issue slots away)
      double b0[4], b1[4];
#pragma unroll
      for (int s = 0; s < 3; ++s) { b0[s] = p0[4 * s]; b1[s] = p1[4 * s]; }
#pragma unroll
      for (int s = 0; s < NSc; ++s) {
        if (s + 3 < NSc) { b0[(s + 3) & 3] = p0[4 * (s + 3)]; b1[(s + 3) & 3] = p1[4 * (s + 3)]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < RBW; ++i) {
          const int ee = s * RBW + i;
          const double av = ring[ee % PF];
          {
            const int en = ee + PF, sn = en / RBW;            // (compile-time after unrolling)
            ring[ee % PF] = en < NSc * RBW ? cbase[en % RBW][sn / 8][(sn % 8) * 64 + lane]
                                           : nbase[en % RBW][(sn - NSc) * 64 + lane];
          }
          acc[i][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b0[s & 3], acc[i][0], 0, 0, 0);
          acc[i][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b1[s & 3], acc[i][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s == 2) { build_offsets(nch); __builtin_amdgcn_sched_barrier(0); }
        if (s == 5) { build_loads(0, Zn); __builtin_amdgcn_sched_barrier(0); }
        if (s == 8) { build_stores(0, Pn); __builtin_amdgcn_sched_barrier(0); }
        if (s == 11) { build_loads(1, Zn); __builtin_amdgcn_sched_barrier(0); }
        if (s == 14) { build_stores(1, Pn); __builtin_amdgcn_sched_barrier(0); }
        if (s == 17) { build_loads(2, Zn); __builtin_amdgcn_sched_barrier(0); }
        if (s == 20) { build_stores(2, Pn); __builtin_amdgcn_sched_barrier(0); }
      }
      if (ch == 0) {
        if (single) {                         // this chunk's hooks read z~(t + G) from Zs[cur ^ 1]; Zs[cur] (this tile's rows,
          store_z(t + 2 * G, Zs + cur * T * ZS);   // last read during the previous tile) takes the tile after next
          load_z(t + 3 * G);
        } else {                              // z~ of the next tile: read from the last chunk on
          store_z(t + G, Zs + (cur ^ 1) * T * ZS);
          load_z(t + 2 * G);
        }
      }
      pbuf ^= 1;
      WSTAMP(0);
      wg_sync();
      WSTAMP(1);
    }
    cur ^= 1;

    // ---- softmax over k on the accumulators -------------------------------------------------
    __builtin_amdgcn_s_setprio(2);
#pragma unroll
    for (int i = 0; i < RBW; ++i)
      if (wave + NW * i >= K16) {               // (scalar) no such row block: 16 padding components
        acc[i][0] = d4{kPadLogDensity, kPadLogDensity, kPadLogDensity, kPadLogDensity};
        acc[i][1] = acc[i][0];
      }
    double m[2], ssum[2], ssel[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      double v = fmax(fmax(acc[0][c][0], acc[0][c][1]), fmax(acc[0][c][2], acc[0][c][3]));
#pragma unroll
      for (int i = 1; i < RBW; ++i) v = fmax(v, fmax(fmax(acc[i][c][0], acc[i][c][1]), fmax(acc[i][c][2], acc[i][c][3])));
      v = fmax(v, __shfl_xor(v, 16));
      v = fmax(v, __shfl_xor(v, 32));
      m[c] = v;
    }
    if (q == 0) { red[wave * T + j] = m[0]; red[wave * T + 16 + j] = m[1]; }
    if (a.logp) {
#pragma unroll
      for (int i = 0; i < RBW; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int k = 16 * (wave + NW * i) + q + 4 * r;
            const int64_t n = n0 + 16 * c + j;
            if (k < K && n < N) a.logp[(int64_t)k * N + n] = acc[i][c][r];
          }
    }
    WSTAMP(2);
    wg_sync();
    WSTAMP(3);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      double v = red[16 * c + j];
#pragma unroll
      for (int w = 1; w < NW; ++w) v = fmax(v, red[w * T + 16 * c + j]);
      m[c] = v;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      double se = 0.0, sl = 0.0;
#pragma unroll
      for (int i = 0; i < RBW; ++i) {
        double e4[4], t4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double l = acc[i][c][r];
          const int k = 16 * (wave + NW * i) + q + 4 * r;
          e4[r] = exp_nonpos(l - m[c], etab);
          t4[r] = want_sel ? e4[r] * ((k < K && l > kOffLogDensity) ? l : 0.0) : 0.0;   // switched-off / padding components: 0 * l := 0
          acc[i][c][r] = e4[r];
        }
        se += (e4[0] + e4[1]) + (e4[2] + e4[3]);
        sl += (t4[0] + t4[1]) + (t4[2] + t4[3]);
      }
      se += __shfl_xor(se, 16); se += __shfl_xor(se, 32);
      sl += __shfl_xor(sl, 16); sl += __shfl_xor(sl, 32);
      ssum[c] = se; ssel[c] = sl;
    }
    if (q == 0) {
      red[(NW + wave) * T + j] = ssum[0]; red[(NW + wave) * T + 16 + j] = ssum[1];
      red[(2 * NW + wave) * T + j] = ssel[0]; red[(2 * NW + wave) * T + 16 + j] = ssel[1];
    }
    WSTAMP(4);
    if (gibbs) {
      // label draw (mimo/utils/stats.py:10-17), scale-invariant form of normalise_tile: label = #{k : u E_K > E_k} on the
      // UNNORMALISED cumulative sums E_k.  Component order = row block, then q + 4 r inside it: the row-block sums go
      // through LDS, inside a block the four q lanes scan each r and carry the rows before it.
      double bsum[RBW][2];
#pragma unroll
      for (int i = 0; i < RBW; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          double v = (acc[i][c][0] + acc[i][c][1]) + (acc[i][c][2] + acc[i][c][3]);
          v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
          bsum[i][c] = v;
          if (q == 0 && wave + NW * i < 16) rbs[(wave + NW * i) * T + 16 * c + j] = v;
        }
      if (wave == 0 && lane < T) {
        const int64_t n = n0 + lane;
        ured[lane] = a.u ? (n < N ? a.u[n] : 0.0) : philox_uniform(a.seed, (uint64_t)(a.row0 + n), a.sweep);
      }
    }
    wg_sync();
    WSTAMP(5);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      double v = red[NW * T + 16 * c + j], u = red[2 * NW * T + 16 * c + j];
#pragma unroll
      for (int w = 1; w < NW; ++w) { v += red[(NW + w) * T + 16 * c + j]; u += red[(2 * NW + w) * T + 16 * c + j]; }
      ssum[c] = v; ssel[c] = u;
    }
    if (gibbs) {
      int cnt[2] = {0, 0};
      double ctot[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        // totals in row-block order (this is E_K); prefix of the blocks in front of each of this wave's blocks
        double run = 0.0, base[RBW];
        for (int rb = 0; rb < K16; ++rb) {
#pragma unroll
          for (int i = 0; i < RBW; ++i) base[i] = rb == wave + NW * i ? run : base[i];
          run += rbs[rb * T + 16 * c + j];
        }
        ctot[c] = run;
        const double tl = ured[16 * c + j] * run;
#pragma unroll
        for (int i = 0; i < RBW; ++i) {
          if (wave + NW * i < K16) {
            double carry = base[i];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              // inclusive scan over the four q lanes and their total from ONE butterfly: s01 = the pair sum (e0 + e1 or
              // e2 + e3), t2 = the other pair's; prefix = (q odd ? s01 : e) + (q >= 2 ? t2 : 0) — for q = 2 that is
              // (e0 + e1) + e2, the order of the reference's cumulative sum
              const double e = acc[i][c][r];
              const double s01 = e + __shfl_xor(e, 16);
              const double t2 = __shfl_xor(s01, 32);
              const double p = ((q & 1) ? s01 : e) + ((q & 2) ? t2 : 0.0);
              cnt[c] += tl > carry + p ? 1 : 0;
              carry += s01 + t2;
            }
          }
        }
        cnt[c] += __shfl_xor(cnt[c], 16);
        cnt[c] += __shfl_xor(cnt[c], 32);
      }
      if (q == 0) { cred[wave * T + j] = cnt[0]; cred[wave * T + 16 + j] = cnt[1]; }
      wg_sync();
      if (wave == 0 && q == 0) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int64_t n = n0 + 16 * c + j;
          if (n < N) {
            int tot = cred[16 * c + j];
#pragma unroll
            for (int w = 1; w < NW; ++w) tot += cred[w * T + 16 * c + j];
            if (a.labels) a.labels[n] = tot < K ? tot : K - 1;
            const double lse = m[c] + log(ctot[c]);
            sc_lse += lse;
            sc_rl += ssel[c] / ctot[c];
            if (a.lse) a.lse[n] = lse;
          }
        }
      }
    } else {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int64_t n = n0 + 16 * c + j;
      const bool valid = n < N;
      double inv = __builtin_amdgcn_rcp(ssum[c]);
      inv = fma(fma(-ssum[c], inv, 1.0), inv, inv);
      inv = fma(fma(-ssum[c], inv, 1.0), inv, inv);
      if (wave == 0 && q == 0 && valid) {         // one lane per datum owns the scalars
        const double lse = m[c] + log(ssum[c]);
        sc_lse += lse;
        sc_rl += ssel[c] * inv;
        if (a.lse) a.lse[n] = lse;
      }
      if (a.resp && valid) {
#pragma unroll
        for (int i = 0; i < RBW; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int k = 16 * (wave + NW * i) + q + 4 * r;
            if (k < K) a.resp[(int64_t)k * N + n] = acc[i][c][r] * inv;
          }
      }
    }
    }
    __builtin_amdgcn_s_setprio(0);
    WSTAMP(6);
    // (the next writes of red[] follow the first chunk barrier of the next tile)
  }
#ifdef MIMO_STAMPS
  if (a.stamps && lane == 0)
    for (int i = 0; i < 8; ++i) a.stamps[(size_t)2 * 8192 * 32 + ((size_t)blockIdx.x * 4 + wave) * 8 + i] = st_sum[i];
#endif
  sc_lse = wave_sum(sc_lse);
  sc_rl = wave_sum(sc_rl);
  if (tid == 0) {
    const size_t pstride = (size_t)K16 * 16 * a.F16_total + 4;
    double* Ps = a.partials + (size_t)blockIdx.x * pstride + (size_t)K16 * 16 * a.F16_total;
    Ps[0] = sc_lse; Ps[1] = sc_rl; Ps[2] = want_sel ? 1.0 : 0.0; Ps[3] = 0.0;     // [2] > 0 after the reduction: the split is valid
  }
}

// ------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------
static int wide_min_d() {
  // measured at N=2e6 against the round-1 pair (which these shapes ran on): Dz=16, K=128 4.64 -> 3.48 ms per VI pass,
  // Dz=16, K=200 12.5 -> 8.7, Dz=14, K=100 3.41 -> 3.01, Dz=12, K=128 2.89 -> 2.74, Dz=13, K=256 8.35 -> 8.47
  static const int v = [] { const char* e = getenv("MIMO_WIDE_MIN_D"); return e ? atoi(e) : 10; }();   // tuning knob
  return v;
}
bool wide_stats_covers(int K16, int D) {
  static const bool on = [] { const char* e = getenv("MIMO_WIDE_STATS"); return !e || atoi(e) != 0; }();   // tuning knob
  // (K <= 32 — two row blocks, four column parts of four blocks — was tried: Dz=32, K=32 2.0 against 1.94 ms of the split tile
  //  kernels, Dz=24, K=24 1.33 / 1.20, Dz=20, K=30 0.71 / 0.92: no clear gain, not instantiated)
  return on && D >= wide_min_d() && D <= kMaxD && K16 >= 3 && K16 <= 16;
}
// column parts of the eight waves: 8 / (waves that share the row blocks)
// 128 < K <= 192: four waves x three row blocks x two column parts of four blocks (8 column blocks per launch, no padding
// row blocks) where that saves a launch over eight waves x two row blocks x six blocks (Dz=32: 5 launches of 1.73 ms against
// 6 of 1.65 ms; Dz=16, 10 column blocks: 2 launches either way, and the second form has no spill: 1.58 against 1.77 ms)
static bool wide_three(int K16, int ncb_total) { return K16 > 8 && K16 <= 12 && (ncb_total + 7) / 8 < (ncb_total + 5) / 6; }
// column parts: K16 <= 4: four waves x one row block; 5 .. 8: eight waves; 9 .. 16: eight waves x two row blocks, or the above
static int wide_cp(int K16, int ncb_total) { return wide_three(K16, ncb_total) ? 2 : K16 > 4 ? 1 : 2; }
// column blocks per launch: as few launches as 12 blocks per wave allow, of equal size
int wide_stats_group_ncb(int K16, int ncb_total) {
  // accumulator blocks per wave: 12 (K <= 128), 8 (K <= 64: two feature tiles of 32 x (16 * 16 + 2) doubles in LDS), 6 x 2 row
  // blocks (K <= 256)
  const int cp = wide_cp(K16, ncb_total), cap = (wide_three(K16, ncb_total) ? 4 : K16 > 8 ? 6 : K16 > 4 ? kWideNCBL : 8) * cp;
  const int launches = (ncb_total + cap - 1) / cap;
  return (ncb_total + launches - 1) / launches;
}
static int wide_ncbl(int K16, int ncb, int ncb_total) {          // accumulator blocks per wave of the instantiation for this launch
  const int cp = wide_cp(K16, ncb_total), need = (ncb + cp - 1) / cp;
  return need <= 4 ? 4 : need <= 6 ? 6 : need <= 8 ? 8 : need <= 10 ? 10 : 12;
}
size_t wide_stats_lds_bytes(int D, int K16, int ncb, int ncb_total) {
  (void)D;
  const int ZS = 35, cp = wide_cp(K16, ncb_total), RS = 16 * wide_ncbl(K16, ncb, ncb_total) * cp + 2;
  return sizeof(double) * ((size_t)2 * kTile * ZS + (size_t)2 * kTile * RS) + sizeof(uint32_t) * 16 * (size_t)ncb;
}
hipError_t launch_wide_stats(const KernelArgs& a, int grid, hipStream_t stream) {
  typedef void (*fn_t)(const KernelArgs);
  const int ncb = a.F16 / 16, nt = a.F16_total / 16, cp = wide_cp(a.K16, nt);
  if (ncb < 1 || ncb > (cp == 1 ? kWideNCBL : 16) || a.D * kTile > 2 * kWideWG) return hipErrorInvalidValue;
  fn_t fn = nullptr;
  if (wide_three(a.K16, nt)) {
    if (ncb > 8) return hipErrorInvalidValue;
    fn = wide_stats_kernel<4, 4, 3>;
  } else if (a.K16 > 8) {
    if (ncb > 6) return hipErrorInvalidValue;
    fn = wide_ncbl(a.K16, ncb, nt) == 4 ? wide_stats_kernel<8, 4, 2> : wide_stats_kernel<8, 6, 2>;
  } else switch (wide_ncbl(a.K16, ncb, nt)) {
    case 4: fn = cp == 1 ? wide_stats_kernel<8, 4> : wide_stats_kernel<4, 4>; break;
    case 6: fn = cp == 1 ? wide_stats_kernel<8, 6> : wide_stats_kernel<4, 6>; break;
    case 8: fn = cp == 1 ? wide_stats_kernel<8, 8> : wide_stats_kernel<4, 8>; break;
    case 10: fn = cp == 1 ? wide_stats_kernel<8, 10> : nullptr; break;
    default: fn = cp == 1 ? wide_stats_kernel<8, 12> : nullptr; break;
  }
  const size_t lds = wide_stats_lds_bytes(a.D, a.K16, ncb, nt);
  if (!fn || lds > 160 * 1024) return hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(kWideWG), lds, stream, a);
  return hipGetLastError();
}


bool wide_estep_covers(int K16, int D, int F16, int gibbs) {
  static const bool on = [] { const char* e = getenv("MIMO_WIDE_ESTEP"); return !e || atoi(e) != 0; }();   // tuning knob
  // (reduced feature maps — diagonal, linear — stay with the chunked kernel: F16 >= 80 is the full map from Dz = 10 on)
  // K16 = 4 (one row block per wave, all four busy), N = 2e6, against the chunked kernel: softmax pass Dz=32, K=64 2.77 / 2.86 ms,
  // Dz=24, K=56 1.92 / 2.22, Dz=20, K=60 1.57 / 1.63 — here; label draw 2.82 / 2.77, 2.03 / 2.15, 1.64 / 1.53 — stays there.
  // K16 = 3: Dz=28, K=40 2.33 against 2.72 ms.
  static const int min_k16 = [] { const char* e = getenv("MIMO_WIDE_ESTEP_MIN_K16"); return e ? atoi(e) : 5; }();   // tuning knob
  const bool k_ok = K16 >= min_k16 || K16 == 3 || (K16 == 4 && !gibbs);
  return on && D >= wide_min_d() && D <= kMaxD && k_ok && K16 <= 16 && F16 >= 80;
}
size_t wide_estep_lds_bytes(int D, int F16) {
  const int ZS = kWideEstepZS, CF = kWideEstepCF, nchunk = (F16 + CF - 1) / CF;
  (void)D;
  return sizeof(double) * ((size_t)2 * kTile * ZS + (size_t)2 * kTile * (CF + 2) + 3 * 4 * kTile + 16 * kTile + kTile + 64) +
         sizeof(int) * 4 * kTile + sizeof(uint32_t) * (size_t)nchunk * CF;
}
hipError_t launch_wide_estep(const KernelArgs& a, int grid, hipStream_t stream) {
  typedef void (*fn_t)(const KernelArgs);
  if (a.K16 < 1 || a.K16 > 16 || a.D * kTile > 4 * kWG || a.F16 < 16) return hipErrorInvalidValue;
  fn_t fn = a.K16 > 12 ? wide_estep_kernel<4> : a.K16 > 8 ? wide_estep_kernel<3> : a.K16 > 4 ? wide_estep_kernel<2> : wide_estep_kernel<1>;
  const size_t lds = wide_estep_lds_bytes(a.D, a.F16);
  if (lds > 80 * 1024) return hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(kWG), lds, stream, a);
  return hipGetLastError();
}

}  // namespace mimo
