// gfx950 (MI355X / CDNA4) kernels of the mimo E-step / sufficient-statistics engine.
//
// Formulation.  With z~ = [z, 1] and the F = (D+1)(D+2)/2 features phi_f(z) = z~_a z~_b (a <= b),
// the canonical log-density  l[k,n] = c_k + b_k.z_n - 1/2 z_n' W_k z_n  is the dense product
//
//        L (K x N) = Theta (K x F) . Phi' (F x N)
//
// and the responsibility-weighted sufficient statistics (n_k, sum r z, sum r z z') are
//
//        S (K x F) = R (K x N) . Phi (N x F) .
//
// Both run on the float64 matrix cores (v_mfma_f64_16x16x4_f64) and execute 2F flops per
// (datum, component) each, which is the symmetric-minimal algorithmic count; the softmax /
// categorical draw over k sits between the two products and never leaves LDS.  A workgroup
// (4 wavefronts) walks a grid-stride sequence of 32-row tiles:
//
//   1. coalesced read of the (32, D) tile of Z into LDS (the next tile is prefetched into registers)
//   2. feature tile Phi (32 x F16) built once in LDS, shared by all waves and both products
//   3. L tile = Theta.Phi': Theta's MFMA A-operand slices stream from the L2-resident operand image
//      through a small register ring (wave w owns component row-blocks w, w+4, ...)
//   4. per-datum normalisation over k (8 lanes per datum): softmax -> r, or inverse-CDF
//      categorical draw (host uniforms or in-kernel Philox4x32-10) -> label
//   5. S += R.Phi accumulated in registers across all tiles of the workgroup
//
// Per-workgroup partial S blocks are written once at the end and summed in a fixed order by
// reduce_partials (no float atomics: results are run-to-run identical).
//
// The float64 VALU shares its pipe with the float64 MFMA on gfx950 (tools/f64_rates.hip), so the
// non-matrix phases are written for few, independent f64 instructions; see DESIGN.md section 4.
//
// LDS bank layout (MI355X_MICROARCH.md, LDS): the feature tile row stride RS = F16 + 1 doubles (odd) is
// conflict-free under the ds_read2_b64 / ds_write2_b64 forms hipcc emits for both MFMA operand patterns:
// step 3 reads Phi[row j][4s + q], step 5 reads Phi[row 8q + s][16cb + j].
//
// Reference behaviour reproduced (paths relative to the reference root):
//   mimo/distributions/gaussian.py:510-521, bayesian.py:287-301, lingauss.py:330-345,
//   bayesian.py:933-947 (log-density tables); mimo/mixtures/gmm.py:72-75,256-259 (softmax);
//   mimo/utils/stats.py:8-21 (inverse-CDF draw: label = #{k : u*cum_K > cum_k});
//   gaussian.py:491-502, lingauss.py:306-322, categorical.py:35-43 (statistics).
#include "mimo_tile.h"
#include "mimo_extra.h"
#include <cstdio>
#include <cstdlib>

#include <math.h>
#include <type_traits>
#include <utility>

namespace mimo {

double philox_uniform_host(uint64_t seed, uint64_t row, uint64_t sweep) {
  return philox_uniform(seed, row, sweep);
}

// Wave priority of the matrix phases.  Two workgroups share a CU; at equal priority they fall into lock step (both in
// their MFMA phase, then both in their latency-bound build phase with the pipe idle).  MIMO_ASYM_PRIO = 1 / 2 runs the
// matrix phases of one of the two at priority 1 (1: upper half of the grid, 2: odd workgroups), so the other one's
// MFMAs fill exactly the gaps.  Scheduling only: results do not depend on it.
#ifndef MIMO_ASYM_PRIO
#define MIMO_ASYM_PRIO 0
#endif
#define MFMA_PRIO()                                                                        \
  do {                                                                                    \
    if (MIMO_ASYM_PRIO && prio_hi) __builtin_amdgcn_s_setprio(1);                         \
    else __builtin_amdgcn_s_setprio(0);                                                   \
  } while (0)

// Start-up stagger (experiment): the second workgroup of a CU starts MIMO_STAGGER x ~1024 cycles late, so that its matrix
// phases fall into the other one's latency-bound phases.
#ifndef MIMO_STAGGER
#define MIMO_STAGGER 0
#endif
#define STAGGER_START()                                                                    \
  do {                                                                                    \
    if (MIMO_STAGGER > 0 && 2 * blockIdx.x >= gridDim.x)                                  \
      for (int i_ = 0; i_ < MIMO_STAGGER; ++i_) __builtin_amdgcn_s_sleep(16);             \
  } while (0)

// ------------------------------------------------------------------------------------------
// Fused tile kernel.  NCB: 16-wide feature column blocks (F16 = 16*NCB); RBW: component
// row-blocks (16 components each) per wavefront; SRC: where the weight tile comes from.
// ------------------------------------------------------------------------------------------
#ifndef MIMO_RBW4_ESTEP_WGS
#define MIMO_RBW4_ESTEP_WGS 2   // workgroups per CU the RBW = 4 E-step kernels with NCB <= 3 are compiled for
#endif
template <int NCB, int RBW, int MODE, int DS = 0, int SPLIT = 0>
__global__ __launch_bounds__(kWG, (RBW == 1 ? 2 : (MODE <= kGeneric && RBW == 2) ? 2 : (MODE <= kGeneric && NCB <= 3) ? MIMO_RBW4_ESTEP_WGS
                                   : (MODE > kGeneric && RBW * NCB <= 12) ? 2 : 1))
void fused_kernel(const KernelArgs a) {
  constexpr int SRC = MODE == kModeWeights ? kSrcWeights : MODE == kModeLabels ? kSrcLabels : kSrcEstep;
  // flags fold to constants in the two fast modes
  const bool gibbs = MODE == kFastVI ? false : MODE == kFastGibbs ? true : a.gibbs != 0;
  double* const out_logp = MODE == kGeneric ? a.logp : nullptr;
  double* const out_resp = MODE == kGeneric ? a.resp : nullptr;
  double* const out_lse = MODE == kGeneric ? a.lse : nullptr;
  const bool do_stats = (MODE == kFastVI || MODE == kFastGibbs) ? true : a.do_stats != 0;
  constexpr int NSI = 4 * NCB;  // contraction steps of 4 features in the Theta image (row-block stride)
  // steps actually needed: with Dz known at compile time the zero-padded tail of the last column block is skipped
  constexpr int NS = DS > 0 ? ((DS + 1) * (DS + 2) / 2 + 3) / 4 : NSI;
  constexpr int T = kTile;

  extern __shared__ __align__(16) unsigned char smem[];
  double* Zs = reinterpret_cast<double*>(smem);  // [T][ZS]   z~ rows (z, 1, 0)
  double* Ph = Zs + T * a.ZS;                    // [T][RS]   feature tile
  double* Lt = Ph + T * a.RS;                    // [T][LS]   l -> e -> r per (row, component)
  double* red = Lt + T * a.LS;                   // [16]      block-reduction scratch
  double* etab = red + 16;                       // [64]      2^(j/64) for exp_nonpos
  // Dz >= 14 (E-step modes, one row block per wave): the 2048-entry exp table (mimo_tile.h); else 64 entries
  constexpr bool E2K = DS >= 14 && RBW == 1 && MODE <= kGeneric;
  uint8_t* fe = reinterpret_cast<uint8_t*>(etab + (E2K ? kExpTab : 64));  // [F16][2]
  int* labs = reinterpret_cast<int*>(red);       // [32] labels of the tile's rows (red is idle until the epilogue)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: wave-uniform branches and SGPR bases
  const bool prio_hi = MIMO_ASYM_PRIO == 2 ? (blockIdx.x & 1) != 0 : 2 * blockIdx.x >= gridDim.x;   // (scalar)
  const int j = lane & 15, q = lane >> 4;
  const int D = a.D, K = a.K, K16 = a.K16, F16 = a.F16;
  const int ZS = a.ZS, RS = a.RS, LS = a.LS;
  const int Kpad = K16 * 16;
  const int64_t N = a.N;

  // statistics modes may cover only the column blocks [cb0, cb0 + NCB) of a larger feature set
  const uint8_t* featp = a.feat + (SRC == kSrcEstep ? 0 : 32 * a.cb0);
  for (int e = tid; e < F16 * 2; e += kWG) fe[e] = featp[e];
  if constexpr (E2K) {
    for (int e = tid; e < kExpTab; e += kWG) etab[e] = exp_tab_entry_c(e);
  } else {
    if (tid < 64) etab[tid] = exp2((double)tid * (1.0 / 64.0));
  }

  // Theta in MFMA A-operand layout: lane (i = lane&15, kk = lane>>4) of slice s of row-block rb holds
  // Theta[16 rb + i][4 s + kk]; the image is prepared on the host so each slice is one coalesced
  // 512-byte read.  The NS*RBW slices a wave consumes per tile are streamed from the L2-resident
  // image through an 8-deep register ring (element e = s*RBW + i), refilled as soon as a slot is
  // consumed and wrapping into the next tile, instead of pinning 2*NS*RBW VGPRs.
  constexpr int NE = NS * RBW;
  constexpr int PF = NE < 8 ? NE : 8;
  // The stream is padded to NEP (a multiple of the ring depth) with dummy refills, so that element e of
  // EVERY tile sits in slot e % PF (otherwise the wrap-around prefetch lands one tile's first slices in
  // the wrong slots whenever NE % PF != 0).
  constexpr int NEP = (NE + PF - 1) / PF * PF;
  double ring[PF];
  // scalar base of this wave's first row block + per-lane element; slice offsets are immediates
  gptr_t thw = (gptr_t)(a.theta + (size_t)wave * NSI * 64);
  // element order of the stream: pass h (RP row blocks at a time), contraction step s, row block i2 of the pass
  constexpr int RP = RBW == 3 ? 3 : RBW >= 2 ? 2 : 1;      // (three row blocks: one pass of three)
  auto theta_slice = [&](int e) -> double {
    if (e >= NE) e = 0;   // padding element: any valid slice, never consumed
    const int h = e / (NS * RP), rem = e % (NS * RP);
    const int s = rem / RP, i = h * RP + rem % RP;
#ifdef MIMO_WHATIF_THETA_L1       // what-if build (wrong results): every slice from the same two cache lines
    return thw[(e & 1) * 64 + lane];
#endif
    return thw[(4 * i * NSI + s) * 64 + lane];
  };
  const gptr_t thw0 = thw;
  if constexpr (SRC == kSrcEstep) {
#pragma unroll
    for (int e = 0; e < PF; ++e) ring[e] = theta_slice(e);
  }

  d4 sacc[RBW][NCB];
#pragma unroll
  for (int i = 0; i < RBW; ++i)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) sacc[i][cb] = d4{0.0, 0.0, 0.0, 0.0};

  double sc_lse = 0.0, sc_rl = 0.0, sc_prod = 1.0;
  int prod_tiles = 0;
  PhiloxBatch pbatch;
  // K <= 16 (one row block): row-block ownership would leave waves 1..3 without matrix work — a K = 16 sweep cost
  // what a K = 64 sweep costs.  Instead the two column groups (rows 0-15 / 16-31) of the L tile go to waves 0 and 1,
  // and the feature column blocks of the statistics to all four waves (wave w: blocks w, w + 4, w + 8, held in
  // sacc[0][0..NWS)).  Separate instantiations (SPLIT, launched for K <= 16 and Dz >= 7): with the two paths in one
  // kernel the C2 instantiation lost 9 % to register allocation.
  // SPLIT = 2 is the same idea for 16 < K <= 32: row block wave & 1, column group / column-block parity wave >> 1.
  static_assert(SPLIT == 0 || (RBW == 1 && SPLIT <= 2), "SPLIT = number of row blocks shared by the four waves");
  constexpr bool is_split = SPLIT != 0;
  constexpr int NRB = SPLIT ? SPLIT : 1, WPR = 4 / NRB;       // row blocks, waves per row block
  constexpr int NWS = (NCB + WPR - 1) / WPR;
  const int srb = wave % NRB, sidx = wave / NRB;              // (scalar) row block and rank of this wave within it

  // Z tile staging: every thread owns up to ZPT elements of the (T, D) tile; the NEXT tile is
  // fetched into registers while the current one is processed, so the HBM latency is off the
  // critical path (T*D <= 512 for the fused E-step modes, Dz <= 16; <= 1024 for the statistics modes).
  constexpr int ZPT = (SRC == kSrcEstep && DS > 0) ? 2 : 4;   // DS = 0: table-driven features, Dz up to 32
  int zoff[ZPT];
#pragma unroll
  for (int i = 0; i < ZPT; ++i) {
    const int e = tid + kWG * i;
    const int pt = e / D;
    zoff[i] = e < T * D ? pt * ZS + (e - pt * D) : -1;
  }
  double zr[ZPT];
  auto load_z = [&](int64_t t) {
    const int64_t base = t * T * D, total = N * D;
#pragma unroll
    for (int i = 0; i < ZPT; ++i) {
      const int64_t g = base + tid + kWG * i;
      zr[i] = (zoff[i] >= 0 && g < total) ? a.Z[g] : 0.0;
    }
  };
  auto store_z = [&](int64_t t) {
#pragma unroll
    for (int i = 0; i < ZPT; ++i)
      if (zoff[i] >= 0) Zs[zoff[i]] = zr[i];
    if (tid < T) {
      Zs[tid * ZS + D] = (t * T + tid) < N ? 1.0 : 0.0;  // rows past N contribute nothing
      Zs[tid * ZS + D + 1] = 0.0;                        // padded features read this slot
    }
  };
  load_z(blockIdx.x);
  store_z(blockIdx.x);
  load_z((int64_t)blockIdx.x + gridDim.x);

  // feature build: thread (row = tid & 31, g = tid >> 5) produces the 2*NCB consecutive features
  // [g*2*NCB, (g+1)*2*NCB); their (a,b) byte pairs are NCB consecutive 32-bit words of the table.
  const int frow = tid & (T - 1), fgrp = tid >> 5;
  wg_sync();  // feature table and exp table are in LDS
  uint32_t w[NCB];
  if constexpr (DS == 0) {
#pragma unroll
    for (int jj = 0; jj < NCB; ++jj) w[jj] = reinterpret_cast<const uint32_t*>(fe)[fgrp * NCB + jj];
  }

  STAGGER_START();
#ifdef MIMO_STAMPS
  unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
#ifdef MIMO_STAMPS
  int st_it = 0;
#endif
  for (int64_t t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
    const int64_t n0 = t * T;
    wg_sync();  // z~ tile of this step is in LDS; the previous tile's readers are done
    STAMP(0);
#ifdef MIMO_STAMPS
    // absolute phase boundaries of iterations 8..23 of the two workgroups of one CU (0 and gridDim.x / 2), wave 0
    const bool st_trace = a.stamps && wave == 0 && lane == 0 && (blockIdx.x == 0 || 2 * blockIdx.x == gridDim.x) &&
                          st_it >= 8 && st_it < 24;
    unsigned long long* st_tr = a.stamps + (size_t)3 * 8192 * 32 + (blockIdx.x ? 64 : 0) + 4 * (st_it - 8);
    if (st_trace) st_tr[0] = st_prev;
    ++st_it;
#endif

    // ---- 2. feature tile (+ externally supplied weights) --------------------------------
    // VALU phases run at raised priority: the f64 VALU shares its pipe with the co-resident
    // workgroup's MFMA stream, and at equal priority a dependent VALU chain gets one issue slot
    // per 64-cycle MFMA.
    __builtin_amdgcn_s_setprio(2);
    if constexpr (DS > 0) {
      // compile-time feature map: the z~ row goes to registers once, every product has static operands
      double zl[DS + 2];
      const double* zrow = Zs + frow * ZS;
#pragma unroll
      for (int d = 0; d <= DS; ++d) zl[d] = zrow[d];
      zl[DS + 1] = 0.0;
      double* prow = Ph + frow * RS + (lane >> 5);
      constexpr int FW = 4 * NCB;
      using Seq = std::make_integer_sequence<int, FW / 2>;
      switch (wave) {   // scalar: no divergence
        case 0: build_features_static<DS, FW, 0>(zl, prow, (lane >> 5) != 0, Seq{}); break;
        case 1: build_features_static<DS, FW, 1>(zl, prow, (lane >> 5) != 0, Seq{}); break;
        case 2: build_features_static<DS, FW, 2>(zl, prow, (lane >> 5) != 0, Seq{}); break;
        default: build_features_static<DS, FW, 3>(zl, prow, (lane >> 5) != 0, Seq{}); break;
      }
    } else {
      const double* zrow = Zs + frow * ZS;
      double* prow = Ph + frow * RS + fgrp * (2 * NCB);
#pragma unroll
      for (int jj = 0; jj < NCB; ++jj) {
        uint32_t wj = w[jj];
        asm volatile("" : "+v"(wj));  // opaque per tile: keeps the 4*NCB derived LDS addresses out of registers
        const double za0 = zrow[wj & 255u], zb0 = zrow[(wj >> 8) & 255u];
        const double za1 = zrow[(wj >> 16) & 255u], zb1 = zrow[wj >> 24];
        prow[2 * jj] = za0 * zb0;
        prow[2 * jj + 1] = za1 * zb1;
      }
    }
    if constexpr (SRC == kSrcWeights) {
      // weight tile (32 rows x Kpad) from the K-major table: 8 * RBW independent loads per thread, issued
      // back to back (a runtime-bound loop would expose one HBM round trip per iteration)
      const int pt = tid & (T - 1), kq = tid >> 5;
      const int64_t n = n0 + pt;
      double wv[8 * RBW];
#pragma unroll
      for (int it = 0; it < 8 * RBW; ++it) {
        const int k = kq + 8 * it;
        wv[it] = (k < K && n < N) ? a.resp[(int64_t)k * N + n] : 0.0;
      }
#pragma unroll
      for (int it = 0; it < 8 * RBW; ++it) {
        const int k = kq + 8 * it;
        if (k < Kpad) Lt[pt * LS + k] = wv[it];
      }
    } else if constexpr (SRC == kSrcLabels) {
      if (tid < T) labs[tid] = (n0 + tid) < N ? a.labels[n0 + tid] : -1;   // one-hot operand is built on the fly
    }
    STAMP(1);
    MFMA_PRIO();
    wg_sync();
    STAMP(2);
#ifdef MIMO_STAMPS
    if (st_trace) st_tr[1] = st_prev;
#endif

    if constexpr (SRC == kSrcEstep) {
      // ---- 3. L tile = Theta . Phi' ------------------------------------------------------
      // B operand: lane (kk = q, col = j) holds Phi[row 16 g + j][4 s + q].
      // C/D layout of v_mfma_f64_16x16x4_f64: reg r of lane (q, j) = row q + 4 r, col j.
      if constexpr (is_split) {
        if (sidx < 2) {                     // column group sidx of row block srb
          gptr_t th = (gptr_t)(a.theta + (size_t)srb * NSI * 64);
          asm volatile("" : "+s"(th));
          const double* p = Ph + (16 * sidx + j) * RS + q;
          d4 acc = d4{0.0, 0.0, 0.0, 0.0};
          constexpr int PD = NS < 8 ? NS : 8;       // Theta slices in flight (straight from L2, no cross-tile ring)
          double tr[PD], bq[3];
#pragma unroll
          for (int e = 0; e < PD; ++e) tr[e] = th[e * 64 + lane];
          bq[0] = p[0];
          if (NS > 1) bq[1] = p[4];
#pragma unroll
          for (int s2 = 0; s2 < NS; ++s2) {
            if (s2 + 2 < NS) bq[(s2 + 2) % 3] = p[4 * (s2 + 2)];
            __builtin_amdgcn_sched_barrier(0);
            const double av = tr[s2 % PD];
            if (s2 + PD < NS) tr[s2 % PD] = th[(s2 + PD) * 64 + lane];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bq[s2 % 3], acc, 0, 0, 0);
          }
          int lw_off = (16 * sidx + j) * LS + 16 * srb + q;
          asm volatile("" : "+v"(lw_off));
          double* lw = Lt + lw_off;
#pragma unroll
          for (int r = 0; r < 4; ++r) lw[4 * r] = acc[r];
        }
      } else if (wave < K16) {   // wave-uniform (scalar) test: this wave owns at least row block `wave`
        thw = thw0;
        asm volatile("" : "+s"(thw));  // opaque per tile: slice addresses = scalar base + immediates, not 2*NE hoisted VGPRs
        const double* p0 = Ph + j * RS + q;
        const double* p1 = Ph + (16 + j) * RS + q;
#pragma unroll
        for (int h = 0; h < RBW / RP; ++h) {       // RP row blocks per pass bound the live accumulators
          if (RBW == 1 || wave + 4 * h * RP < K16) {   // scalar: the pass has at least one live row block
            d4 acc[RP][2];
#pragma unroll
            for (int i2 = 0; i2 < RP; ++i2) { acc[i2][0] = d4{0.0, 0.0, 0.0, 0.0}; acc[i2][1] = d4{0.0, 0.0, 0.0, 0.0}; }
            // B operands are read two contraction steps ahead of their MFMAs (an LDS read is ~2-3 MFMA
            // issue slots away); sched_barriers keep hipcc from sinking the reads back to their use
            double bq0[3], bq1[3];
            bq0[0] = p0[0]; bq1[0] = p1[0];
            if (NS > 1) { bq0[1] = p0[4]; bq1[1] = p1[4]; }
#pragma unroll
            for (int s = 0; s < NS; ++s) {     // straight-line: a dead second row block of the pass
              if (s + 2 < NS) { bq0[(s + 2) % 3] = p0[4 * (s + 2)]; bq1[(s + 2) % 3] = p1[4 * (s + 2)]; }
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int i2 = 0; i2 < RP; ++i2) {   // multiplies zero-padded Theta (results never stored)
                const int e = h * NS * RP + s * RP + i2;
                const double av = ring[e % PF];
                ring[e % PF] = theta_slice((e + PF) % NEP);   // wraps into the next tile's first slices
                acc[i2][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bq0[s % 3], acc[i2][0], 0, 0, 0);
                acc[i2][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bq1[s % 3], acc[i2][1], 0, 0, 0);
              }
            }
            int lw_off = j * LS + 16 * wave + q;
            asm volatile("" : "+v"(lw_off));
            double* lw0 = Lt + lw_off;
            double* lw1 = lw0 + 16 * LS;
#pragma unroll
            for (int i2 = 0; i2 < RP; ++i2) {
              const int i = h * RP + i2;
              if (RBW == 1 || wave + 4 * i < K16) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  lw0[64 * i + 4 * r] = acc[i2][0][r];
                  lw1[64 * i + 4 * r] = acc[i2][1][r];
                }
              }
            }
          } else {
#pragma unroll
            for (int ee = 0; ee < NS * RP; ++ee) {   // dead pass: keep the Theta ring in step
              const int e = h * NS * RP + ee;
              ring[e % PF] = theta_slice((e + PF) % NEP);
            }
          }
        }
#pragma unroll
        for (int e = NE; e < NEP; ++e) ring[e % PF] = theta_slice((e + PF) % NEP);   // stream padding
      }
      STAMP(3);
      wg_sync();
      STAMP(4);

      // ---- 4. normalise over k: 8 lanes per datum, 2*K16 consecutive components per lane ----------
      __builtin_amdgcn_s_setprio(2);
      if constexpr (RBW == 1)
        normalise_tile<RBW, MODE, E2K>(a, Lt, LS, etab, K, K16, N, n0, wave, lane, gibbs, out_logp, out_resp, out_lse,
                                  sc_lse, sc_rl, sc_prod, labs, pbatch, (int64_t)gridDim.x * T);
      else
        normalise_tile_chunked<RBW, MODE>(a, Lt, LS, etab, K, K16, N, n0, wave, lane, gibbs, out_logp, out_resp,
                                          out_lse, sc_lse, sc_rl, sc_prod, labs, pbatch, (int64_t)gridDim.x * T);
      if constexpr (MODE != kGeneric) {
        if (++prod_tiles == 64) {   // K^64 <= 256^64 = 2^512 stays inside the float64 range
          sc_lse += log(sc_prod);
          sc_prod = 1.0;
          prod_tiles = 0;
        }
      }
      STAMP(5);
      MFMA_PRIO();
      wg_sync();
      STAMP(6);
    }
#ifdef MIMO_STAMPS
    if (st_trace) st_tr[2] = st_prev;
#endif

    // ---- 5. S += R . Phi ----------------------------------------------------------------
    // step s contracts the 4 rows {s, s+8, s+16, s+24}: A lane (i = j, kk = q) = R[8q+s][16rb+j],
    // B lane (kk = q, col = j) = Phi[8q+s][16cb+j].
    if constexpr (is_split) {
     if (do_stats) {
      int lt_off = 8 * q * LS + 16 * srb + j, ph_off = 8 * q * RS + j;
      asm volatile("" : "+v"(lt_off), "+v"(ph_off));
      const double* ltq = Lt + lt_off;
      const double* phq = Ph + ph_off;
      int cbo[NWS];      // scalar: column offsets of this wave's blocks (a block past NCB repeats the last one; never stored)
#pragma unroll
      for (int i = 0; i < NWS; ++i) cbo[i] = 16 * (sidx + WPR * i < NCB ? sidx + WPR * i : NCB - 1);
      auto stats_split = [&](auto lab_c) {
        constexpr bool LAB = decltype(lab_c)::value;
        double avq[2], bvq[2][NWS];
        auto fetch = [&](int s2, int slot) {
          const double* pb = phq + s2 * RS;
          if constexpr (LAB) avq[slot] = labs[8 * q + s2] == 16 * srb + j ? 1.0 : 0.0;
          else avq[slot] = ltq[s2 * LS];
#pragma unroll
          for (int i = 0; i < NWS; ++i) bvq[slot][i] = pb[cbo[i]];
        };
        fetch(0, 0);
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
          if (s2 + 1 < 8) fetch(s2 + 1, (s2 + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < NWS; ++i)
            sacc[0][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(avq[s2 & 1], bvq[s2 & 1][i], sacc[0][i], 0, 0, 0);
        }
      };
      if constexpr (MODE == kFastGibbs || MODE == kModeLabels) stats_split(std::true_type{});
      else if constexpr (MODE == kGeneric) { if (gibbs) stats_split(std::true_type{}); else stats_split(std::false_type{}); }
      else stats_split(std::false_type{});
     }
    } else if (do_stats && wave < K16) {
      // per-tile opaque bases: every operand address below is base + s * stride + immediate, instead of
      // 8 * (RBW + 1) loop-invariant addresses the compiler would otherwise pin in (and spill from) VGPRs
      // (the OFFSETS are made opaque, not the pointers: a pointer that went through an asm loses its LDS
      // address space and every read would become a flat_load with a full vmcnt/lgkmcnt drain)
      int lt_off = 8 * q * LS + 16 * wave + j, ph_off = 8 * q * RS + j;
      asm volatile("" : "+v"(lt_off), "+v"(ph_off));
      const double* ltq = Lt + lt_off;
      const double* phq = Ph + ph_off;
      // straight-line body for NACT active row blocks of this wave (no branch inside: LDS reads pipeline
      // across the 8 steps); the B operand of a step is read once and reused by all NACT row blocks
      // LAB: the weights are one-hot(labels) -> A operand = (label of row 8q+s == this lane's component)
      auto stats_body = [&](auto nact_c, auto lab_c) {
        constexpr int NACT = decltype(nact_c)::value;
        constexpr bool LAB = decltype(lab_c)::value;
        const int comp0 = 16 * wave + j;
        // operands of step s+1 are read before the MFMAs of step s are issued
        double avq[2][NACT], bvq[2][NCB];
        auto fetch = [&](int s, int slot) {
          const double* lts = ltq + s * LS;
          const double* pb = phq + s * RS;
          int lab = 0;
          if constexpr (LAB) lab = labs[8 * q + s];
#pragma unroll
          for (int i = 0; i < NACT; ++i)
            avq[slot][i] = LAB ? (lab == comp0 + 64 * i ? 1.0 : 0.0) : lts[64 * i];
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) bvq[slot][cb] = pb[16 * cb];
        };
        fetch(0, 0);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          if (s + 1 < 8) fetch(s + 1, (s + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < NACT; ++i)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
              sacc[i][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(avq[s & 1][i], bvq[s & 1][cb], sacc[i][cb], 0, 0, 0);
        }
      };
      auto stats_nact = [&](auto lab_c) {
        const int nact = (K16 - wave + 3) >> 2;   // scalar: row blocks wave, wave+4, ... below K16
        if constexpr (RBW == 1) {
          stats_body(std::integral_constant<int, 1>{}, lab_c);
        } else if constexpr (RBW == 2) {
          if (nact >= 2) stats_body(std::integral_constant<int, 2>{}, lab_c);
          else stats_body(std::integral_constant<int, 1>{}, lab_c);
        } else if constexpr (RBW == 3) {
          if (nact >= 3) stats_body(std::integral_constant<int, 3>{}, lab_c);
          else if (nact == 2) stats_body(std::integral_constant<int, 2>{}, lab_c);
          else stats_body(std::integral_constant<int, 1>{}, lab_c);
        } else {
          if (nact >= 4) stats_body(std::integral_constant<int, 4>{}, lab_c);
          else if (nact == 3) stats_body(std::integral_constant<int, 3>{}, lab_c);
          else if (nact == 2) stats_body(std::integral_constant<int, 2>{}, lab_c);
          else stats_body(std::integral_constant<int, 1>{}, lab_c);
        }
      };
      // Hard labels, K > 16: the weight tile is one-hot, so row block rb only receives the rows whose label
      // falls into it — on average 32/K16 of the 32.  Instead of contracting all 8 steps against a mostly
      // zero A operand, the members of each row block are compacted (wave ballot -> scalar bit scan) and
      // contracted 4 at a time: ~1 step per row block instead of 8 (K = 256: 12 MFMAs per wave-tile
      // instead of 96).  Every row is a member of exactly one row block, so the sums are the same sums.
      auto stats_sparse = [&]() {
        const int rbl = labs[lane & 31] >> 4;          // row block of row (lane & 31); -1 for rows beyond N
        const int shift = 8 * q;
#pragma unroll
        for (int i = 0; i < RBW; ++i) {
          const int rb = wave + 4 * i;
          if (rb < K16) {
            uint32_t m = (uint32_t)__ballot(rbl == rb);   // lanes 32..63 repeat lanes 0..31: low half only
            while (m) {
              uint32_t packed = 0;                         // row indices of the next (up to) 4 members, 0xFF = none
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const uint32_t d = m ? (uint32_t)__builtin_ctz(m) : 0xFFu;
                m &= m - 1;
                packed |= d << (8 * e);
              }
              const uint32_t mine = (packed >> shift) & 0xFFu;   // lane (kk = q) contracts member q
              const bool have = mine != 0xFFu;
              const int row_d = have ? (int)mine : 0;
              const double av = (have && (labs[row_d] & 15) == j) ? 1.0 : 0.0;
              const double* pb = Ph + row_d * RS + j;
#pragma unroll
              for (int cb = 0; cb < NCB; ++cb)
                sacc[i][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, pb[16 * cb], sacc[i][cb], 0, 0, 0);
            }
          }
        }
      };
      auto stats_labels = [&]() {
        if (K16 >= 2) stats_sparse();
        else stats_nact(std::true_type{});
      };
      if constexpr (MODE == kFastGibbs || MODE == kModeLabels) stats_labels();
      else if constexpr (MODE == kGeneric) { if (gibbs) stats_labels(); else stats_nact(std::false_type{}); }
      else stats_nact(std::false_type{});
    }

    // ---- 1'. stage the next tile's z~ rows (Zs was last read before the barrier after step 2)
    store_z(t + gridDim.x);
    load_z(t + 2 * (int64_t)gridDim.x);
    STAMP(7);
#ifdef MIMO_STAMPS
    if (st_trace) st_tr[3] = st_prev;
#endif
  }
#ifdef MIMO_STAMPS
  if (a.stamps && lane == 0)
    for (int i = 0; i < 8; ++i) a.stamps[((size_t)blockIdx.x * 4 + wave) * 8 + i] = st_sum[i];
#endif

  // ---- per-workgroup partials ------------------------------------------------------------
  const int FT = a.F16_total;   // row stride of the partial block (= F16 unless this launch is one column group)
  const size_t pstride = (size_t)Kpad * FT + 4;
  double* P = a.partials + (size_t)blockIdx.x * pstride + (SRC == kSrcEstep ? 0 : 16 * a.cb0);
  if constexpr (is_split) {
#pragma unroll
    for (int i = 0; i < NWS; ++i) {
      const int cb = sidx + WPR * i;
      if (cb < NCB) {
#pragma unroll
        for (int r = 0; r < 4; ++r) P[(size_t)(16 * srb + q + 4 * r) * FT + 16 * cb + j] = sacc[0][i][r];
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < RBW; ++i) {
      const int rb = wave + 4 * i;
      if (rb < K16) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            P[(size_t)(16 * rb + q + 4 * r) * FT + 16 * cb + j] = sacc[i][cb][r];
      }
    }
  }
  if constexpr (MODE == kFastVI || MODE == kFastGibbs) sc_lse += log(sc_prod);
  sc_lse = wave_sum(sc_lse);
  sc_rl = wave_sum(sc_rl);
  wg_sync();
  if (lane == 0) { red[2 * wave] = sc_lse; red[2 * wave + 1] = sc_rl; }
  wg_sync();
  if (tid == 0 && a.write_scalars) {
    double* Ps = a.partials + (size_t)blockIdx.x * pstride + (size_t)Kpad * FT;
    Ps[0] = (red[0] + red[2]) + (red[4] + red[6]);
    Ps[1] = (red[1] + red[3]) + (red[5] + red[7]);
    Ps[2] = MODE == kGeneric ? 1.0 : 0.0;   // > 0 after the reduction: split is valid
    Ps[3] = 0.0;
  }
}


// ------------------------------------------------------------------------------------------
// Chunked E-step (no statistics) for shapes the fused kernel cannot hold: the F16 features are
// processed in chunks of 16*kChunkNCB; per chunk the feature tile is rebuilt in LDS and the L tile
// accumulates in registers.  Writes the responsibility table / labels (+ optional logp, lse) to
// HBM; the statistics then come from fused_kernel<.., kModeWeights / kModeLabels> per column group.
// ------------------------------------------------------------------------------------------
template <int RBW, int SPLIT = 0>
__global__ __launch_bounds__(kWG, (RBW <= 2 ? 2 : 1)) void estep_chunked_kernel(const KernelArgs a) {
  // SPLIT (K <= 32, see fused_kernel): NRB = SPLIT row blocks shared by the four waves — wave w works on row block
  // w % NRB and on ONE column group (w / NRB) of the L tile instead of both; waves with w / NRB >= 2 have no part.
  static_assert(SPLIT == 0 || (RBW == 1 && SPLIT <= 2), "SPLIT = number of row blocks shared by the four waves");
  constexpr int NRB = SPLIT ? SPLIT : 1;
  constexpr int T = kTile;
  constexpr int NCBc = kChunkNCB, CF = 16 * NCBc, NSc = CF / 4;
  #ifndef MIMO_CHUNK_RING
#define MIMO_CHUNK_RING 6
#endif
  constexpr int RP = RBW >= 2 ? 2 : 1, NPASS = RBW / RP, LE = NSc * RP, PF = MIMO_CHUNK_RING;
  static_assert(LE % PF == 0, "ring slots must line up across blocks");
  extern __shared__ __align__(16) unsigned char smem[];
  double* Zs = reinterpret_cast<double*>(smem);
  double* Ph = Zs + T * a.ZS;
  double* Lt = Ph + T * a.RS;
  double* red = Lt + T * a.LS;
  double* etab = red + 16;
  uint8_t* fe = reinterpret_cast<uint8_t*>(etab + 64);   // [nchunk * CF][2]
  int* labs = reinterpret_cast<int*>(red);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool prio_hi = MIMO_ASYM_PRIO == 2 ? (blockIdx.x & 1) != 0 : 2 * blockIdx.x >= gridDim.x;   // (scalar)
  const int j = lane & 15, q = lane >> 4;
  const int D = a.D, K = a.K, K16 = a.K16, F16 = a.F16;
  const int ZS = a.ZS, RS = a.RS, LS = a.LS;
  const int nchunk = (F16 + CF - 1) / CF, NSP = chunked_ns_pad(F16);   // Theta image is zero-padded to NSP steps per row block
  const int NB = nchunk * NPASS;                                 // (chunk, pass) blocks per tile
  const int64_t N = a.N;
  const bool gibbs = a.gibbs != 0;
  if (tid < 64) etab[tid] = exp2((double)tid * (1.0 / 64.0));
  for (int e = tid; e < nchunk * CF * 2; e += kWG) fe[e] = e < F16 * 2 ? a.feat[e] : (uint8_t)(D + 1);
  double sc_lse = 0.0, sc_rl = 0.0, sc_prod = 1.0;
  PhiloxBatch pbatch;
  const int frow = tid & (T - 1), fgrp = tid >> 5;   // feature build: 8 groups x 2*NCBc features per chunk

  // Theta stream: block bl = (chunk, pass) consumes LE slices in the order (step s, row block i2 of the
  // pass); a 6-deep register ring prefetches across block and tile boundaries.
  const int srb = SPLIT ? wave % NRB : wave, sidx = SPLIT ? wave / NRB : 0;   // (scalar)
  const bool mfma_wave = SPLIT ? sidx < 2 : wave < K16;
  gptr_t thw = (gptr_t)(a.theta + (size_t)srb * NSP * 64 + lane);
  auto block_base = [&](int bl) {
    const int ch = bl / NPASS, h = bl - ch * NPASS;
    return thw + ((size_t)(4 * h * RP) * NSP + (size_t)ch * NSc) * 64;
  };
#ifdef MIMO_WHATIF_THETA_L1       // what-if build (wrong results): every slice from the same two cache lines
  auto slice = [&](gptr_t base, int ee) { return thw[(ee & 1) * 64]; };
#else
  auto slice = [&](gptr_t base, int ee) { return base[((size_t)(4 * (ee % RP)) * NSP + ee / RP) * 64]; };
#endif
  double ring[PF];
  if (mfma_wave) {
#pragma unroll
    for (int e = 0; e < PF; ++e) ring[e] = slice(block_base(0), e);
  }

  STAGGER_START();
  // Z tile staging as in fused_kernel: the NEXT tile's rows travel in registers while this one is processed
  // (T*D <= 1024 elements for Dz <= 32: 4 per thread)
  constexpr int ZPT = 4;
  int zoff[ZPT];
#pragma unroll
  for (int i = 0; i < ZPT; ++i) {
    const int e = tid + kWG * i;
    const int pt = e / D;
    zoff[i] = e < T * D ? pt * ZS + (e - pt * D) : -1;
  }
  double zr[ZPT];
  auto load_z = [&](int64_t t) {
    const int64_t base = t * T * D, total = N * D;
#pragma unroll
    for (int i = 0; i < ZPT; ++i) {
      const int64_t g = base + tid + kWG * i;
      zr[i] = (zoff[i] >= 0 && g < total) ? a.Z[g] : 0.0;
    }
  };
  auto store_z = [&](int64_t t) {
#pragma unroll
    for (int i = 0; i < ZPT; ++i)
      if (zoff[i] >= 0) Zs[zoff[i]] = zr[i];
    if (tid < T) {
      Zs[tid * ZS + D] = (t * T + tid) < N ? 1.0 : 0.0;
      Zs[tid * ZS + D + 1] = 0.0;
    }
  };
  load_z(blockIdx.x);
#ifdef MIMO_STAMPS
  unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
  for (int64_t t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
    const int64_t n0 = t * T;
    wg_sync();
    STAMP(0);
    store_z(t);                                   // fetched while the previous tile was processed
    load_z(t + gridDim.x);
    d4 acc[RBW][2];
#pragma unroll
    for (int i = 0; i < RBW; ++i) { acc[i][0] = d4{0.0, 0.0, 0.0, 0.0}; acc[i][1] = d4{0.0, 0.0, 0.0, 0.0}; }

    STAMP(1);
    for (int ch = 0; ch < nchunk; ++ch) {
      wg_sync();   // z~ rows visible / previous chunk's MFMA reads of Ph are done
      STAMP(2);
      if (MIMO_ASYM_PRIO) __builtin_amdgcn_s_setprio(2);
      {
        // 2*NCBc features of one row per thread: the (a, b) byte pairs arrive as NCBc dwords, then all operands of a
        // half are read before its products are stored (reads and writes alias for the compiler: interleaved, every
        // feature would wait out its own three LDS round trips)
        const double* zrow = Zs + frow * ZS;
        double* prow = Ph + frow * RS + fgrp * (2 * NCBc);
        const uint32_t* ftw = reinterpret_cast<const uint32_t*>(fe + 2 * (ch * CF + fgrp * 2 * NCBc));
        uint32_t wds[NCBc];
#pragma unroll
        for (int i = 0; i < NCBc; ++i) wds[i] = ftw[i];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          double za[NCBc], zb[NCBc];
#pragma unroll
          for (int i = 0; i < NCBc; ++i) {
            const int jj = h * NCBc + i;
            const uint32_t w2 = wds[jj >> 1] >> (16 * (jj & 1));
            za[i] = zrow[w2 & 255u];
            zb[i] = zrow[(w2 >> 8) & 255u];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < NCBc; ++i) prow[h * NCBc + i] = za[i] * zb[i];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      STAMP(3);
      if (MIMO_ASYM_PRIO) MFMA_PRIO();
      wg_sync();
      STAMP(4);
      if constexpr (SPLIT != 0) {
        if (mfma_wave) {     // RBW = 1: one pass, LE = NSc slices per chunk, one column group
          const double* p = Ph + (16 * sidx + j) * RS + q;
          gptr_t base = block_base(ch);
          gptr_t nbase = block_base(ch + 1 == NB ? 0 : ch + 1);
#pragma unroll
          for (int ee = 0; ee < LE; ++ee) {
            const double av = ring[ee % PF];
            ring[ee % PF] = ee + PF < LE ? slice(base, ee + PF) : slice(nbase, ee + PF - LE);
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, p[4 * ee], acc[0][0], 0, 0, 0);
          }
        }
      } else if (wave < K16) {
        const double* p0 = Ph + j * RS + q;
        const double* p1 = Ph + (16 + j) * RS + q;
#pragma unroll
        for (int h = 0; h < NPASS; ++h) {
          const int bl = ch * NPASS + h;
          gptr_t base = block_base(bl);
          gptr_t nbase = block_base(bl + 1 == NB ? 0 : bl + 1);
          if (RBW == 1 || wave + 4 * h * RP < K16) {
            double b0 = 0.0, b1 = 0.0;
#pragma unroll
            for (int ee = 0; ee < LE; ++ee) {
              const int s = ee / RP, i2 = ee % RP;
              const double av = ring[ee % PF];
              ring[ee % PF] = ee + PF < LE ? slice(base, ee + PF) : slice(nbase, ee + PF - LE);
              if (i2 == 0) { b0 = p0[4 * s]; b1 = p1[4 * s]; }
              acc[h * RP + i2][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b0, acc[h * RP + i2][0], 0, 0, 0);
              acc[h * RP + i2][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b1, acc[h * RP + i2][1], 0, 0, 0);
            }
          } else {
#pragma unroll
            for (int ee = 0; ee < LE; ++ee)   // dead pass: keep the ring in step
              ring[ee % PF] = ee + PF < LE ? slice(base, ee + PF) : slice(nbase, ee + PF - LE);
          }
        }
      }
      STAMP(5);
    }
    if constexpr (SPLIT != 0) {
      if (mfma_wave) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Lt[(16 * sidx + j) * LS + 16 * srb + q + 4 * r] = acc[0][0][r];
      }
    } else {
#pragma unroll
      for (int i = 0; i < RBW; ++i) {
        const int rb = wave + 4 * i;
        if (rb < K16) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            Lt[j * LS + 16 * rb + q + 4 * r] = acc[i][0][r];
            Lt[(16 + j) * LS + 16 * rb + q + 4 * r] = acc[i][1][r];
          }
        }
      }
    }
    wg_sync();
    STAMP(6);
    __builtin_amdgcn_s_setprio(2);
    if constexpr (RBW == 1)
      normalise_tile<RBW, kGeneric>(a, Lt, LS, etab, K, K16, N, n0, wave, lane, gibbs, a.logp, a.resp, a.lse,
                                    sc_lse, sc_rl, sc_prod, labs, pbatch, (int64_t)gridDim.x * T);
    else
      normalise_tile_chunked<RBW, kGeneric>(a, Lt, LS, etab, K, K16, N, n0, wave, lane, gibbs, a.logp, a.resp,
                                            a.lse, sc_lse, sc_rl, sc_prod, labs, pbatch, (int64_t)gridDim.x * T);
    MFMA_PRIO();
    STAMP(7);
  }
#ifdef MIMO_STAMPS
  if (a.stamps && lane == 0)
    for (int i = 0; i < 8; ++i) a.stamps[(size_t)8192 * 32 + ((size_t)blockIdx.x * 4 + wave) * 8 + i] = st_sum[i];
#endif
  sc_lse = wave_sum(sc_lse);
  sc_rl = wave_sum(sc_rl);
  wg_sync();
  if (lane == 0) { red[2 * wave] = sc_lse; red[2 * wave + 1] = sc_rl; }
  wg_sync();
  if (tid == 0) {
    double* Ps = a.partials + (size_t)blockIdx.x * ((size_t)K16 * 16 * a.F16_total + 4) + (size_t)K16 * 16 * a.F16_total;
    Ps[0] = (red[0] + red[2]) + (red[4] + red[6]);
    Ps[1] = (red[1] + red[3]) + (red[5] + red[7]);
    Ps[2] = 1.0;
    Ps[3] = 0.0;
  }
}

// out[e] = sum_g partials[g][e] in a fixed order (4 interleaved chains, then a fixed tree).

// ------------------------------------------------------------------------------------------
// Posterior-predictive mixture moments (mimo/mixtures/ilr.py:339-372,374-430): for every row x_n
//   l_k   = c_k + b_k.x - 1/2 x'W_k x                 log gating mean + log basis predictive
//   w_k   = softmax_k l_k
//   m_k   = M_k x~,  cs_k = 1 + x~'Q_k x~,  V_k = cs_k C_k      expert predictive mean / covariance
//   average: mu = sum_k w_k m_k,  covar = sum_k w_k (V_k + m_k m_k') - mu mu'
//   mode   : k* = argmax_k w_k,   mu = m_k*,  covar = V_k*
//   nlpd (y given) = -logsumexp_k [ log N(y; m_k, (P_k / cs_k)^-1) + log(w_k + tiny) ]
// One thread per row, the K loop streams the per-component blocks through scalar loads (uniform
// addresses), x~ lives in a thread-private LDS column (runtime dx), the dy + dy^2 accumulators in
// registers (DY is a template parameter).  Online softmax: one pass over k for the moments.
// Inference-time N is small; this is VALU work by design.
// ------------------------------------------------------------------------------------------
template <int DY>
__global__ __launch_bounds__(256) void predict_kernel(const PredictArgs a) {
  extern __shared__ double xs[];   // [dc][256]
  const int tid = threadIdx.x;
  const int64_t n = (int64_t)blockIdx.x * 256 + tid;
  const bool valid = n < a.N;
  const int dx = a.dx, dc = a.dc, K = a.K;
  for (int i = 0; i < dx; ++i) xs[i * 256 + tid] = valid ? a.Z[n * dx + i] : 0.0;
  if (dc > dx) xs[dx * 256 + tid] = 1.0;
  auto X = [&](int i) { return xs[i * 256 + tid]; };

  auto gate = [&](int k) {
    const double* t = a.gate + (size_t)k * (1 + dx + dx * dx);
    double l = t[0];
    for (int i = 0; i < dx; ++i) {
      double q = 0.0;
      for (int j = 0; j < dx; ++j) q = fma(t[1 + dx + i * dx + j], X(j), q);
      l = fma(X(i), t[1 + i] - 0.5 * q, l);
    }
    return l;
  };
  auto expert = [&](int k, double (&m)[DY], double& cs) {
    const double* Mk = a.M + (size_t)k * DY * dc;
    const double* Qk = a.Q + (size_t)k * dc * dc;
#pragma unroll
    for (int d = 0; d < DY; ++d) m[d] = 0.0;
    double q = 0.0;
    for (int i = 0; i < dc; ++i) {
      const double xi = X(i);
      double r = 0.0;
      for (int j = 0; j < dc; ++j) r = fma(Qk[i * dc + j], X(j), r);
      q = fma(xi, r, q);
#pragma unroll
      for (int d = 0; d < DY; ++d) m[d] = fma(Mk[d * dc + i], xi, m[d]);
    }
    cs = 1.0 + q;
  };

  double mx = -INFINITY, ssum = 0.0;
  double amu[DY], aS[DY][DY];
#pragma unroll
  for (int d = 0; d < DY; ++d) {
    amu[d] = 0.0;
#pragma unroll
    for (int e = 0; e < DY; ++e) aS[d][e] = 0.0;
  }
  int best = 0;
  for (int k = 0; k < K; ++k) {
    const double l = gate(k);
    if (a.mode == 1) {          // argmax only (first maximum, as np.argmax)
      if (l > mx) { mx = l; best = k; }
      continue;
    }
    if (l > mx) {               // rescale the running sums to the new maximum
      const double sc = exp(mx - l);
      ssum *= sc;
#pragma unroll
      for (int d = 0; d < DY; ++d) {
        amu[d] *= sc;
#pragma unroll
        for (int e = 0; e < DY; ++e) aS[d][e] *= sc;
      }
      mx = l;
    }
    const double w = exp(l - mx);
    double m[DY], cs;
    expert(k, m, cs);
    const double* Ck = a.Cc + (size_t)k * DY * DY;
    ssum += w;
#pragma unroll
    for (int d = 0; d < DY; ++d) {
      amu[d] = fma(w, m[d], amu[d]);
#pragma unroll
      for (int e = 0; e < DY; ++e) aS[d][e] = fma(w, fma(cs, Ck[d * DY + e], m[d] * m[e]), aS[d][e]);
    }
  }
  double lse = 0.0;
  if (a.mode == 1) {
    double m[DY], cs;
    expert(best, m, cs);
    const double* Ck = a.Cc + (size_t)best * DY * DY;
    if (valid) {
#pragma unroll
      for (int d = 0; d < DY; ++d) {
        a.mu[n * DY + d] = m[d];
        if (a.diag) {
          const double v = cs * Ck[d * DY + d];
          a.covar[n * DY + d] = v;
          a.covar[(a.N + n) * DY + d] = sqrt(v);
        } else {
#pragma unroll
          for (int e = 0; e < DY; ++e) a.covar[(n * DY + d) * DY + e] = cs * Ck[d * DY + e];
        }
      }
    }
    if (a.nlpd) {               // the log-normaliser is still needed for the weights inside nlpd
      ssum = 0.0;
      for (int k = 0; k < K; ++k) ssum += exp(gate(k) - mx);
    }
  } else if (valid) {
    const double inv = 1.0 / ssum;
#pragma unroll
    for (int d = 0; d < DY; ++d) amu[d] *= inv;
#pragma unroll
    for (int d = 0; d < DY; ++d) {
      a.mu[n * DY + d] = amu[d];
      if (a.diag) {
        const double v = aS[d][d] * inv - amu[d] * amu[d];
        a.covar[n * DY + d] = v;
        a.covar[(a.N + n) * DY + d] = sqrt(v);
      } else {
#pragma unroll
        for (int e = 0; e < DY; ++e) a.covar[(n * DY + d) * DY + e] = aS[d][e] * inv - amu[d] * amu[e];
      }
    }
  }
  if (a.nlpd) {
    lse = mx + log(ssum);
    double yv[DY];
#pragma unroll
    for (int d = 0; d < DY; ++d) yv[d] = valid ? a.y[n * DY + d] : 0.0;
    double tm = -INFINITY, ts = 0.0;
    for (int k = 0; k < K; ++k) {
      const double w = exp(gate(k) - lse);
      double m[DY], cs;
      expert(k, m, cs);
      const double* Pk = a.P + (size_t)k * DY * DY;
      double q = 0.0;
#pragma unroll
      for (int d = 0; d < DY; ++d) {
        double r = 0.0;
#pragma unroll
        for (int e = 0; e < DY; ++e) r = fma(Pk[d * DY + e], yv[e] - m[e], r);
        q = fma(yv[d] - m[d], r, q);
      }
      const double lpl = -0.5 * q / cs - 0.5 * DY * 1.8378770664093453 + 0.5 * (a.ld[k] - DY * log(cs));
      const double t = lpl + log(w + 2.2250738585072014e-308);
      if (t > tm) { ts = ts * exp(tm - t) + 1.0; tm = t; }
      else ts += exp(t - tm);
    }
    if (valid) a.nlpd[n] = -(tm + log(ts));
  }
}

__global__ void reduce_partials(const double* __restrict__ partials, int G, int64_t stride,
                                double* __restrict__ out) {
  // 64 consecutive elements per workgroup, the G partial blocks split into 4 contiguous slices (one per wave),
  // every slice summed in 4 interleaved chains, the slices combined in a fixed order: the association depends
  // on G only, never on timing.  (One thread per element over all G blocks left 41 workgroups for 256 CUs:
  // 40 us per sweep at C2; this shape: 160 workgroups.)
  __shared__ double part[4][64];
  const int ex = threadIdx.x & 63, gs = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + ex;
  const int g0 = (int)((int64_t)G * gs / 4), g1 = (int)((int64_t)G * (gs + 1) / 4);
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (e < stride) {
    int g = g0;
    for (; g + 3 < g1; g += 4) {
      s0 += partials[(int64_t)g * stride + e];
      s1 += partials[(int64_t)(g + 1) * stride + e];
      s2 += partials[(int64_t)(g + 2) * stride + e];
      s3 += partials[(int64_t)(g + 3) * stride + e];
    }
    for (; g < g1; ++g) s0 += partials[(int64_t)g * stride + e];
  }
  part[gs][ex] = (s0 + s1) + (s2 + s3);
  wg_sync();
  if (gs == 0 && e < stride) out[e] = (part[0][ex] + part[1][ex]) + (part[2][ex] + part[3][ex]);
}

// partial[b] = -sum t log t over this block's grid-stride share of a table; entries that are not
// strictly positive contribute 0 (the reference uses nansum(resp * log(resp)), gmm.py:353-355).
__global__ void table_entropy_partials(const double* __restrict__ t, int64_t count,
                                       double* __restrict__ partial) {
  __shared__ double red[4];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * blockDim.x) {
    const double v = t[i];
    if (v > 0.0) s -= v * log(v);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  wg_sync();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// feature-space block [Kpad][F16] (+4 scalars) -> packed S[K][1 + D + D*D] and scalars[3]
__global__ void unpack_stats(const double* __restrict__ red, const uint8_t* __restrict__ feat,
                             int K, int D, int F, int F16, double* __restrict__ S,
                             double* __restrict__ scalars, int mask_structure) {
  const int Kpad = (K + 15) / 16 * 16;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (S && e < (int64_t)K * F) {
    const int k = (int)(e / F), f = (int)(e - (int64_t)k * F);
    const int aa = feat[2 * f], bb = feat[2 * f + 1];
    double v = red[(int64_t)k * F16 + f];
    // (small-shape kernel under a structure hint: it accumulates the full map, the caller is promised zeros)
    if (mask_structure == 1 && aa != bb && bb != D) v = 0.0;     // diagonal W: no off-diagonal second moments
    if (mask_structure == 2 && bb != D) v = 0.0;                 // linear: n_k and sum r z only
    double* Sk = S + (int64_t)k * (1 + D + D * D);
    if (aa == D) Sk[0] = v;                // (D,D): n_k
    else if (bb == D) Sk[1 + aa] = v;      // (a,D): sum r z_a
    else { Sk[1 + D + aa * D + bb] = v; Sk[1 + D + bb * D + aa] = v; }
  }
  if (scalars && e == 0) {
    const double slse = red[(int64_t)Kpad * F16 + 0], srl = red[(int64_t)Kpad * F16 + 1];
    const bool split = red[(int64_t)Kpad * F16 + 2] > 0.0;
    scalars[0] = slse;                       // sum_n logsumexp_k l
    // (integer selects: the file is built with -fno-honor-nans, under which a floating-point select may
    // fold a NaN arm away)
    const long long nan_bits = 0x7ff8000000000000LL;
    scalars[1] = __longlong_as_double(split ? __double_as_longlong(srl) : nan_bits);          // sum_n sum_k r l
    scalars[2] = __longlong_as_double(split ? __double_as_longlong(slse - srl) : nan_bits);   // -sum r log r  (log r = l - lse, sum_k r = 1)
  }
}

// ------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------
// (src = kSrcEstep at Dz = 14 .. 16 with K <= 64 on the full map: the instantiations with the 2048-entry exp table, fused_kernel E2K)
size_t fused_lds_bytes(const KernelArgs& a, int src) {
  const bool e2k = src == kSrcEstep && !a.diag && a.D >= 14 && a.D <= kMaxFusedD && a.K16 <= 4;
  return sizeof(double) * ((size_t)kTile * (a.ZS + a.RS + a.LS) + 16 + (e2k ? kExpTab : 64)) + (size_t)a.F16 * 2;
}

// row blocks per wave: the fused E-step kernels exist for 1 and 4; the statistics modes and the chunked
// E-step also for 2 (K <= 128), which halves the accumulator registers they pin
static int rbw_for(int K16) { return K16 <= 4 ? 1 : 4; }
static int rbw_stats(int K16) { return K16 <= 4 ? 1 : K16 <= 8 ? 2 : 4; }

static int rbw_stats(int K16);

// K > 128 runs 4 row blocks per wave: 5 column blocks (160 accumulator registers) per launch; 64 < K <= 128: 6; else 10
int stats_group_ncb(int K16) {
  static const int knob = [] { const char* e = getenv("MIMO_STATS_GROUP"); return e ? atoi(e) : 0; }();   // tuning knob
  if (knob >= 1 && knob <= kMaxNCB && rbw_stats(K16) >= 2) return knob;
  // RBW = 2 (64 < K <= 128): six column blocks keep the launch at two workgroups per CU (249 VGPRs; seven or more
  // need a whole CU per workgroup and lose the latency hiding): C5 shape 60.8 -> 56.5 ms, measured 3..10
  return rbw_stats(K16) == 4 ? 5 : rbw_stats(K16) == 2 ? 6 : kMaxNCB;
}

bool fused_covers(int K16, int ncb, int src) {
  if (K16 > 16 || ncb < 1 || ncb > kMaxNCB) return false;
  if (src == kSrcEstep) return K16 <= 4 || ncb <= 4;   // RBW = 4 E-step only for Dz <= 9
  return ncb <= stats_group_ncb(K16);
}

typedef void (*fused_fn)(const KernelArgs);

constexpr int ncb_of(int D) { return ((D + 1) * (D + 2) / 2 + 15) / 16; }

// E-step modes: one instantiation per Dz (compile-time feature map)
template <int D, int RBW>
static fused_fn pick_estep_mode(int mode) {
  switch (mode) {
    case kFastVI: return fused_kernel<ncb_of(D), RBW, kFastVI, D>;
    case kFastGibbs: return fused_kernel<ncb_of(D), RBW, kFastGibbs, D>;
    case kGeneric: return fused_kernel<ncb_of(D), RBW, kGeneric, D>;
  }
  return nullptr;
}

template <int RBW>
static fused_fn pick_estep(int D, int mode) {
  switch (D) {
    case 1: return pick_estep_mode<1, RBW>(mode);
    case 2: return pick_estep_mode<2, RBW>(mode);
    case 3: return pick_estep_mode<3, RBW>(mode);
    case 4: return pick_estep_mode<4, RBW>(mode);
    case 5: return pick_estep_mode<5, RBW>(mode);
    case 6: return pick_estep_mode<6, RBW>(mode);
    case 7: return pick_estep_mode<7, RBW>(mode);
    case 8: return pick_estep_mode<8, RBW>(mode);
    case 9: return pick_estep_mode<9, RBW>(mode);
    default: break;
  }
  if constexpr (RBW == 1) {
    switch (D) {
      case 10: return pick_estep_mode<10, 1>(mode);
      case 11: return pick_estep_mode<11, 1>(mode);
      case 12: return pick_estep_mode<12, 1>(mode);
      case 13: return pick_estep_mode<13, 1>(mode);
      case 14: return pick_estep_mode<14, 1>(mode);
      case 15: return pick_estep_mode<15, 1>(mode);
      case 16: return pick_estep_mode<16, 1>(mode);
    }
  }
  return nullptr;
}

// K <= 32 at Dz >= 7 (three or more feature column blocks): the split work distribution (fused_kernel, SPLIT = K16)
template <int D, int SP>
static fused_fn pick_estep_split_mode(int mode) {
  switch (mode) {
    case kFastVI: return fused_kernel<ncb_of(D), 1, kFastVI, D, SP>;
    case kFastGibbs: return fused_kernel<ncb_of(D), 1, kFastGibbs, D, SP>;
    case kGeneric: return fused_kernel<ncb_of(D), 1, kGeneric, D, SP>;
  }
  return nullptr;
}
template <int SP>
static fused_fn pick_estep_split(int D, int mode) {
  static const int min_d = [] { const char* e = getenv("MIMO_SPLIT_MIN_D"); return e ? atoi(e) : 7; }();   // tuning knob
  if (D < min_d) return nullptr;
  switch (D) {
    case 7: return pick_estep_split_mode<7, SP>(mode);
    case 8: return pick_estep_split_mode<8, SP>(mode);
    case 9: return pick_estep_split_mode<9, SP>(mode);
    case 10: return pick_estep_split_mode<10, SP>(mode);
    case 11: return pick_estep_split_mode<11, SP>(mode);
    case 12: return pick_estep_split_mode<12, SP>(mode);
    case 13: return pick_estep_split_mode<13, SP>(mode);
    case 14: return pick_estep_split_mode<14, SP>(mode);
    case 15: return pick_estep_split_mode<15, SP>(mode);
    case 16: return pick_estep_split_mode<16, SP>(mode);
  }
  return nullptr;
}

// E-step modes over a table-driven feature set (diagonal structure: 2 Dz + 1 features, at most 5 column blocks)
template <int RBW>
static fused_fn pick_estep_table(int ncb, int mode) {
#define MIMO_ESTEP_TABLE_CASE(n) case n: return mode == kFastVI ? fused_kernel<n, RBW, kFastVI, 0> \
    : mode == kFastGibbs ? fused_kernel<n, RBW, kFastGibbs, 0> : fused_kernel<n, RBW, kGeneric, 0>;
  switch (ncb) {
    MIMO_ESTEP_TABLE_CASE(1) MIMO_ESTEP_TABLE_CASE(2) MIMO_ESTEP_TABLE_CASE(3) MIMO_ESTEP_TABLE_CASE(4)
    MIMO_ESTEP_TABLE_CASE(5)
  }
#undef MIMO_ESTEP_TABLE_CASE
  return nullptr;
}
// the same for K <= 32 with three or more column blocks (diagonal structure at Dz >= 16): split distribution
template <int SP>
static fused_fn pick_estep_table_split(int ncb, int mode) {
#define MIMO_ESTEP_TABLE_CASE(n) case n: return mode == kFastVI ? fused_kernel<n, 1, kFastVI, 0, SP> \
    : mode == kFastGibbs ? fused_kernel<n, 1, kFastGibbs, 0, SP> : fused_kernel<n, 1, kGeneric, 0, SP>;
  switch (ncb) { MIMO_ESTEP_TABLE_CASE(3) MIMO_ESTEP_TABLE_CASE(4) MIMO_ESTEP_TABLE_CASE(5) }
#undef MIMO_ESTEP_TABLE_CASE
  return nullptr;
}

// statistics modes: table-driven feature build, one instantiation per column-block count
template <int RBW, int SP = 0>
static fused_fn pick_stats(int ncb, int mode) {
#define MIMO_STATS_CASE(n) case n: return mode == kModeWeights ? fused_kernel<n, RBW, kModeWeights, 0, SP> : fused_kernel<n, RBW, kModeLabels, 0, SP>;
  switch (ncb) {
    MIMO_STATS_CASE(1) MIMO_STATS_CASE(2) MIMO_STATS_CASE(3) MIMO_STATS_CASE(4) MIMO_STATS_CASE(5)
    MIMO_STATS_CASE(6) MIMO_STATS_CASE(7) MIMO_STATS_CASE(8) MIMO_STATS_CASE(9) MIMO_STATS_CASE(10)
  }
#undef MIMO_STATS_CASE
  return nullptr;
}

static fused_fn resolve_fused(const KernelArgs& a, int src) {
  const int ncb = a.F16 / 16;
  int mode = src == kSrcWeights ? kModeWeights : src == kSrcLabels ? kModeLabels : kGeneric;
  if (src == kSrcEstep && a.do_stats && !a.split && !a.logp && !a.resp && !a.lse && (a.gibbs || !a.u))
    mode = a.gibbs ? kFastGibbs : kFastVI;   // (per-row weights of a mean-field pass live in the generic kernels)
  if (!fused_covers(a.K16, ncb, src)) return nullptr;
  if (src == kSrcEstep && a.diag && a.K16 <= 2 && ncb >= 3) {
    static const bool on = [] { const char* e = getenv("MIMO_SPLIT_TABLE"); return !e || atoi(e) != 0; }();   // tuning knob
    if (on) return a.K16 == 1 ? pick_estep_table_split<1>(ncb, mode) : pick_estep_table_split<2>(ncb, mode);
  }
  if (src == kSrcEstep && a.diag)
    return rbw_for(a.K16) == 1 ? pick_estep_table<1>(ncb, mode) : (a.K16 <= 8 ? pick_estep_table<2>(ncb, mode) : pick_estep_table<4>(ncb, mode));
  if (src == kSrcEstep && a.K16 <= 2) {
    if (fused_fn f = a.K16 == 1 ? pick_estep_split<1>(a.D, mode) : pick_estep_split<2>(a.D, mode)) return f;
  }
  if (src == kSrcEstep) {
    // 64 < K <= 128 (Dz <= 9): two row blocks per wave instead of four (half of which would multiply padding)
    static const bool rbw2 = [] { const char* e = getenv("MIMO_ESTEP_RBW2"); return !e || atoi(e) != 0; }();   // tuning knob
    if (rbw2 && a.K16 > 4 && a.K16 <= 8) if (fused_fn f = pick_estep<2>(a.D, mode)) return f;
    if (rbw2 && a.K16 > 8 && a.K16 <= 12) if (fused_fn f = pick_estep<3>(a.D, mode)) return f;     // 128 < K <= 192: three
    return rbw_for(a.K16) == 1 ? pick_estep<1>(a.D, mode) : pick_estep<4>(a.D, mode);
  }
  if (a.K16 <= 2 && ncb >= 3) {   // K <= 32: split distribution of the statistics column blocks (see fused_kernel, SPLIT)
    static const bool on = [] { const char* e = getenv("MIMO_SPLIT_STATS"); return !e || atoi(e) != 0; }();   // tuning knob
    if (on) return a.K16 == 1 ? pick_stats<1, 1>(ncb, mode) : pick_stats<1, 2>(ncb, mode);
  }
  return rbw_stats(a.K16) == 1 ? pick_stats<1>(ncb, mode)
         : rbw_stats(a.K16) == 2 ? pick_stats<2>(ncb, mode) : pick_stats<4>(ncb, mode);
}

// Workgroups resident per CU for the kernel this launch resolves to (registers + LDS), from the runtime's
// occupancy calculator; the persistent grid is sized to exactly that, capped at kMaxWGPerCU (beyond it the
// per-workgroup partial blocks and the tail imbalance cost more than the extra latency hiding returns).
constexpr int kMaxWGPerCU = 4;
int fused_grid(const KernelArgs& a, int num_cu, int src) {
  int per_cu = 0;
  if (fused_fn fn = resolve_fused(a, src)) {
    const size_t lds = fused_lds_bytes(a, src);
    if (lds <= 160 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) == hipSuccess) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(fn), kWG, lds) ==
          hipSuccess)
        per_cu = nb;
    }
    (void)hipGetLastError();
  }
  if (per_cu <= 0) {   // two-stage path (or no answer): registers allow two workgroups unless RBW = 4 with wide tiles
    const bool regs2 = rbw_for(a.K16) == 1 || a.F16 / 16 <= 3 || a.K16 <= 8;
    KernelArgs t = a;
    if (t.F16 / 16 > kChunkNCB) { t.F16 = 16 * kChunkNCB; t.RS = t.F16 + 1; }   // per-launch feature tile
    per_cu = (regs2 && fused_lds_bytes(t, kModeWeights) <= 80 * 1024) ? 2 : 1;
  }
  int cap = kMaxWGPerCU;
  if (const char* e = getenv("MIMO_WG_PER_CU")) cap = atoi(e) > 0 ? atoi(e) : cap;   // tuning knob
  if (per_cu > cap) per_cu = cap;
  if (getenv("MIMO_DEBUG")) fprintf(stderr, "[mimo] fused_grid: K16=%d F16=%d src=%d -> %d workgroups/CU\n", a.K16, a.F16, src, per_cu);
  int64_t g = (int64_t)num_cu * per_cu;
  if (g > a.ntiles) g = a.ntiles;
  if (g < 1) g = 1;
  return (int)g;
}

hipError_t launch_fused(const KernelArgs& a, int src, int grid, hipStream_t stream,
                        bool* unsupported) {
  *unsupported = false;
  fused_fn fn = resolve_fused(a, src);
  if (!fn) { *unsupported = true; return hipSuccess; }
  const size_t lds = fused_lds_bytes(a, src);
  if (lds > 160 * 1024) { *unsupported = true; return hipSuccess; }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(kWG), lds, stream, a);
  return hipGetLastError();
}

size_t chunked_lds_bytes(const KernelArgs& a) {
  const int CF = 16 * kChunkNCB, nchunk = (a.F16 + CF - 1) / CF;
  return sizeof(double) * ((size_t)kTile * (a.ZS + a.RS + a.LS) + 16 + 64) + (size_t)nchunk * CF * 2;
}

hipError_t launch_estep_chunked(const KernelArgs& a, int grid, hipStream_t stream) {
  typedef void (*fn_t)(const KernelArgs);
  static const bool split_on = [] { const char* e = getenv("MIMO_SPLIT_CHUNKED"); return !e || atoi(e) != 0; }();   // tuning knob
  fn_t fn = (split_on && a.K16 == 1) ? estep_chunked_kernel<1, 1> : (split_on && a.K16 == 2) ? estep_chunked_kernel<1, 2>
            : rbw_stats(a.K16) == 1 ? estep_chunked_kernel<1>
            : rbw_stats(a.K16) == 2 ? estep_chunked_kernel<2> : estep_chunked_kernel<4>;
  const size_t lds = chunked_lds_bytes(a);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(kWG), lds, stream, a);
  return hipGetLastError();
}

hipError_t launch_reduce(const double* partials, int G, int64_t stride, double* out,
                         hipStream_t stream) {
  hipLaunchKernelGGL(reduce_partials, dim3((unsigned)((stride + 63) / 64)), dim3(256), 0, stream,
                     partials, G, stride, out);
  return hipGetLastError();
}

hipError_t launch_table_entropy(const double* table, int64_t count, double* partials, int nblocks,
                                double* out, hipStream_t stream) {
  hipLaunchKernelGGL(table_entropy_partials, dim3(nblocks), dim3(256), 0, stream, table, count, partials);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return launch_reduce(partials, nblocks, 1, out, stream);
}

hipError_t launch_unpack(const double* reduced, const uint8_t* feat, int K, int D, int F, int F16,
                         double* S_packed, double* scalars3, hipStream_t stream, int mask_structure) {
  const int bs = 256;
  const int64_t total = (int64_t)K * F;
  if (S_packed && F < feat_count(D)) {   // diagonal structure: the off-diagonal second moments are not computed
    hipError_t e = hipMemsetAsync(S_packed, 0, sizeof(double) * (size_t)K * (1 + D + (size_t)D * D), stream);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(unpack_stats, dim3((unsigned)((total + bs - 1) / bs)), dim3(bs), 0, stream,
                     reduced, feat, K, D, F, F16, S_packed, scalars3, mask_structure);
  return hipGetLastError();
}

hipError_t launch_predict(const PredictArgs& a, hipStream_t stream, bool* unsupported) {
  typedef void (*fn_t)(const PredictArgs);
  static const fn_t table[kMaxPredictDy] = {predict_kernel<1>, predict_kernel<2>, predict_kernel<3>, predict_kernel<4>,
                                            predict_kernel<5>, predict_kernel<6>, predict_kernel<7>, predict_kernel<8>};
  *unsupported = a.dy < 1 || a.dy > kMaxPredictDy || a.dc > kMaxD + 1;
  if (*unsupported) return hipSuccess;
  if (a.N <= 0) return hipSuccess;
  hipError_t reg_err = hipSuccess;
  if (launch_predict_reg(a, stream, &reg_err)) return reg_err;      // narrow inputs (dx <= 8, affine): x~ in registers (mimo_predict.hip)
  fn_t fn = table[a.dy - 1];
  const size_t lds = (size_t)a.dc * 256 * sizeof(double);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, dim3((unsigned)((a.N + 255) / 256)), dim3(256), lds, stream, a);
  return hipGetLastError();
}

}  // namespace mimo
