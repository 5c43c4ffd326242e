// Wide shapes (Dz > 16: the two-stage path): statistics S = R . Phi of a K-major weight table.
//
// Why a second kernel next to fused_kernel<.., kModeWeights>: at Dz = 32, K = 128 that one runs as two 4-wave
// workgroups per CU, each staging a 32 x 128 weight tile through LDS and building 6 column blocks of features per
// tile, 6 launches per sweep.  The per-phase trace (tools/stamps_chunked.py) shows the two workgroups of a CU in lock
// step: both wait out the weight tile's HBM round trip and the feature build's LDS chains together (5.1k cycles,
// matrix pipe idle), then share the pipe for their 2 x 96 MFMAs (12.3k) — 66 % of the FP64 rate.  Priorities and a
// start-up stagger do not separate them (a build that overlaps the neighbour's matrix phase gets one VALU issue slot
// per 64-cycle MFMA and falls back into step), so the idle phase itself has to go:
//   * ONE 8-wave workgroup per CU; wave w owns row block w % RBN and every CP-th column block (CP = 8 / RBN) of the
//     launch: 12 accumulator blocks per wave, 12 (RBN = 8) or 24 (RBN = 4) column blocks per launch — the table is
//     read 3 x (K = 128, Dz = 32) instead of 6 x, Z four times instead of seven;
//   * the weights never touch LDS: lane (j, q) of the wave that owns row block rb loads the 8 consecutive rows
//     8q .. 8q+7 of component 16 rb + j — 64 contiguous bytes — and register s IS the MFMA A operand of contraction
//     step s (rows s, s+8, s+16, s+24); the next tile's 8 registers are in flight during this tile's matrix phase;
//   * z rows travel two tiles ahead in registers (double-buffered z tile), so the build phase is LDS-only: 2 barriers
//     per tile, no global latency on the critical path.
// Partial blocks, feature table and reduction are those of the tile kernels (mimo_kernels.hip).
#include "mimo_device.h"
#include "mimo_extra.h"

#include <type_traits>

namespace mimo {

constexpr int kWideWG = 512;       // 8 wavefronts
constexpr int kWideNCBL = 12;      // accumulator column blocks per wave, at most

typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));   // 16-byte load of an 8-byte aligned address

template <int RBN, int NCBL>
__global__ __launch_bounds__(kWideWG) void wide_stats_kernel(const KernelArgs a) {
  constexpr int T = kTile, CP = 8 / RBN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int D = a.D, K = a.K, K16 = a.K16;
  const int ncb = a.F16 / 16;                       // column blocks of this launch: NCBL * CP, or a few less (the
                                                    // surplus blocks are built from offset 0 and never stored)
  const int ZS = (D + 2) | 1;                       // odd: the 32 rows of a column are conflict-free
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
  const int kcomp = 16 * rb + j;
  const double* wrow = a.resp + (int64_t)(kcomp < K ? kcomp : K - 1) * N;
  // A operands of one tile: rows 8q .. 8q+7 of this lane's component, 64 contiguous bytes
  auto load_w = [&](int64_t t, double (&w)[8]) {
    const double* p = wrow + (t < last_tile ? t : last_tile) * T + 8 * q;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const d2u v = *reinterpret_cast<const d2u*>(p + 2 * i);
      w[2 * i] = v[0]; w[2 * i + 1] = v[1];
    }
  };

  d4 acc[NCBL];
#pragma unroll
  for (int i = 0; i < NCBL; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};

  // feature build: thread (row, fgrp) makes feature column fgrp of every column block, in batches of <= 6 features
  // whose operand reads (after matrix step 2b) and product stores (after step 2b + 1) sit BETWEEN the matrix steps of
  // the previous tile: the build's LDS round trips pass under 12 MFMAs each instead of holding all eight waves at a
  // barrier, and one barrier per tile is left.
  const int frow = tid & (T - 1), fgrp = tid >> 5;
  constexpr int NBF = NCBL * CP;                     // features per thread
  constexpr int NBATCH = (NBF + 5) / 6, BF = (NBF + NBATCH - 1) / NBATCH;
  static_assert(NBATCH <= 4, "two matrix steps per batch");
  double* Ph1 = Ph + T * RS;                         // second feature tile

  double wa[8], wb[8];
  // this thread's (a, b) byte offsets stay in registers: a table read inside the matrix phase would put an LDS round
  // trip (s_waitcnt in the in-order instruction stream) in front of the following MFMAs
  uint32_t w2[NBF];
  wg_sync();
#pragma unroll
  for (int i = 0; i < NBF; ++i) w2[i] = i < ncb ? fo[16 * i + fgrp] : 0u;   // columns past ncb: offsets 0, a finite product nobody stores
  double za[BF], zb[BF];
  // batch bt of the feature tile built from the z~ rows at zb_ into the feature tile pb_
  auto build_loads = [&](int bt, const double* zb_) {
    const unsigned char* zrow = reinterpret_cast<const unsigned char*>(zb_ + frow * ZS);
#pragma unroll
    for (int i = 0; i < BF; ++i)
      if (bt * BF + i < NBF) {
        za[i] = *reinterpret_cast<const double*>(zrow + (w2[bt * BF + i] & 0xFFFFu));
        zb[i] = *reinterpret_cast<const double*>(zrow + (w2[bt * BF + i] >> 16));
      }
  };
  auto build_stores = [&](int bt, double* pb_) {
    double* prow = pb_ + frow * RS + fgrp;
#pragma unroll
    for (int i = 0; i < BF; ++i)
      if (bt * BF + i < NBF) prow[16 * (bt * BF + i)] = za[i] * zb[i];
  };

  const int64_t G = gridDim.x;
  int cur = 0;
  // iteration i (tile t): matrix steps on Phi(t) in Ph[cur]; Phi(t + G) built from Zs[cur ^ 1] into Ph[cur ^ 1];
  // z~(t + 2G) staged into Zs[cur]; z(t + 3G) fetched
  auto tile = [&](int64_t t, double (&wc)[8], double (&wn)[8]) {
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
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(wc[s], bv[i], acc[i], 0, 0, 0);
        if (s + 1 < 8) bv[i] = phq[(s + 1) * RS + 16 * CP * i];   // operand of the next step: NCBL MFMAs ahead of its use
      }
      __builtin_amdgcn_sched_barrier(0);      // no hoisting of later steps' reads: 8 x NCBL operands do not fit
      if (s / 2 < NBATCH) {
        if ((s & 1) == 0) build_loads(s / 2, Zn);
        else build_stores(s / 2, Pn);
        __builtin_amdgcn_sched_barrier(0);
      }
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
    for (int bt = 0; bt < NBATCH; ++bt) {
      build_loads(bt, Zs);
      build_stores(bt, Ph);
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
    for (int bt = 0; bt < NBATCH; ++bt) {
      build_loads(bt, Zs);
      build_stores(bt, Ph);
    }
    wg_sync();
    const double* phq = Ph + 8 * q * RS + 16 * cpart + j;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int64_t n = n0 + 8 * q + s;
      const double w = n < N ? wrow[n] : 0.0;
#pragma unroll
      for (int i = 0; i < NCBL; ++i)
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(w, phq[s * RS + 16 * CP * i], acc[i], 0, 0, 0);
    }
  }

  // ---- per-workgroup partial block (layout of the tile kernels)
  const int FT = a.F16_total, Kpad = 16 * K16;
  const size_t pstride = (size_t)Kpad * FT + 4;
  double* P = a.partials + (size_t)blockIdx.x * pstride + 16 * a.cb0;
  if (rb < K16) {
#pragma unroll
    for (int i = 0; i < NCBL; ++i) {
      const int cb = cpart + CP * i;
      if (cb < ncb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = 16 * rb + q + 4 * r;
          P[(size_t)k * FT + 16 * cb + j] = k < K ? acc[i][r] : 0.0;
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
// launch helpers
// ------------------------------------------------------------------------------------------
bool wide_stats_covers(int K16, int D) {
  static const bool on = [] { const char* e = getenv("MIMO_WIDE_STATS"); return !e || atoi(e) != 0; }();   // tuning knob
  return on && D > kMaxFusedD && D <= kMaxD && K16 >= 3 && K16 <= 8;
}
// column blocks per launch: as few launches as 12 blocks per wave allow, of equal size
int wide_stats_group_ncb(int K16, int ncb_total) {
  const int cp = K16 > 4 ? 1 : 2, cap = (K16 > 4 ? kWideNCBL : 8) * cp;    // two feature tiles of 32 x (16 cap + 2) doubles in LDS
  const int launches = (ncb_total + cap - 1) / cap;
  return (ncb_total + launches - 1) / launches;
}
static int wide_ncbl(int K16, int ncb) {          // accumulator blocks per wave of the instantiation for this launch
  const int cp = K16 > 4 ? 1 : 2, need = (ncb + cp - 1) / cp;
  return need <= 4 ? 4 : need <= 6 ? 6 : need <= 8 ? 8 : need <= 10 ? 10 : 12;
}
size_t wide_stats_lds_bytes(int D, int K16, int ncb) {
  const int ZS = (D + 2) | 1, cp = K16 > 4 ? 1 : 2, RS = 16 * wide_ncbl(K16, ncb) * cp + 2;
  return sizeof(double) * ((size_t)2 * kTile * ZS + (size_t)2 * kTile * RS) + sizeof(uint32_t) * 16 * (size_t)ncb;
}
hipError_t launch_wide_stats(const KernelArgs& a, int grid, hipStream_t stream) {
  typedef void (*fn_t)(const KernelArgs);
  const int ncb = a.F16 / 16, cp = a.K16 > 4 ? 1 : 2;
  if (ncb < 1 || ncb > (cp == 1 ? kWideNCBL : 16) || a.D * kTile > 2 * kWideWG) return hipErrorInvalidValue;
  fn_t fn = nullptr;
  switch (wide_ncbl(a.K16, ncb)) {
    case 4: fn = cp == 1 ? wide_stats_kernel<8, 4> : wide_stats_kernel<4, 4>; break;
    case 6: fn = cp == 1 ? wide_stats_kernel<8, 6> : wide_stats_kernel<4, 6>; break;
    case 8: fn = cp == 1 ? wide_stats_kernel<8, 8> : wide_stats_kernel<4, 8>; break;
    case 10: fn = cp == 1 ? wide_stats_kernel<8, 10> : nullptr; break;
    default: fn = cp == 1 ? wide_stats_kernel<8, 12> : nullptr; break;
  }
  const size_t lds = wide_stats_lds_bytes(a.D, a.K16, ncb);
  if (!fn || lds > 160 * 1024) return hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(kWideWG), lds, stream, a);
  return hipGetLastError();
}

}  // namespace mimo
