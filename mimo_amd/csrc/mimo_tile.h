// Device-side pieces of the tile kernels shared by mimo_kernels.hip and mimo_pipe.hip: the Philox batch of the label draw,
// the compile-time feature map, the per-datum normalisation of an LDS-resident L tile, the diagnostic phase stamps.
#pragma once
// Exponential of the tile kernels' normalise phase: 64-entry table + quintic (exp_nonpos, 15 instructions), or — E2K, the
// single-pass E-step kernels at Dz >= 14 — the 2048-entry table with the pre-compensated scale (exp_nonpos_t2048c, 12
// instructions, 16 KB of LDS more per workgroup).  Dz >= 14 runs two workgroups per CU with either table (54 - 64 KB + 16 KB <= 80 KB);
// below, the larger table costs the third workgroup and more than it gives (measured, round 3: C2 kernel 6.71 -> 6.60 ms with it,
// C4 — Dz = 12, three workgroups — 2.13 -> 2.24 ms).
#include "mimo_device.h"

#include <math.h>
#include <type_traits>
#include <utility>

namespace mimo {

#ifndef MIMO_EXP_CHAINS
#define MIMO_EXP_CHAINS 8   // independent exp chains the normalise phase keeps in flight (register-array variant)
#endif

// The 8 lanes that share a datum would all run the same ten Philox rounds (~20 64-bit multiplies + ~40 integer
// instructions on the pipe the f64 MFMAs use).  Instead a wave draws, every 8th tile, the uniforms of its 8 data
// rows for the NEXT 8 tiles of the workgroup's grid-stride walk — lane (pt = lane & 7, j = lane >> 3) holds the
// uniform of row pt of tile t + j * stride — and each tile fetches its value with one lane exchange.  The counter
// is still (global row, sweep): the labels are the same labels.  Used by the K <= 64 kernels only: in the K > 64
// kernels (RBW = 4, already at the 256-register cap) the two extra loop-carried registers pushed other loop-carried
// values into scratch — 3.6 GB of spill writes per launch at C3 for a 4 % gain — so those draw per tile.
struct PhiloxBatch {
  double u = 0.0;
  int used = 8;     // tiles consumed from the batch (8 = empty)
};
__device__ __forceinline__ double philox_for_tile(PhiloxBatch& pb, const KernelArgs& a, const int64_t n, const int lane,
                                                  const int64_t tile_stride_rows) {
  if (pb.used == 8) {     // wave-uniform
    pb.u = philox_uniform(a.seed, (uint64_t)(a.row0 + n + (int64_t)(lane >> 3) * tile_stride_rows), a.sweep);
    pb.used = 0;
  }
  const double uu = __shfl(pb.u, (pb.used << 3) | (lane & 7));
  pb.used += 1;
  return uu;
}

#ifdef MIMO_STAMPS
// diagnostic build: per-wave cycle sums of the phases of the tile loop (never in the shipped library)
#define STAMP(i)                                                                          \
  do {                                                                                    \
    unsigned long long t_;                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    st_sum[i] += t_ - st_prev;                                                            \
    st_prev = t_;                                                                         \
  } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

// ------------------------------------------------------------------------------------------
// Compile-time feature map for the E-step modes (DS = Dz known at compile time): feature f of
// z~ = [z, 1] is z~_a z~_b with (a,b) the f-th pair of the upper triangle, row-major; padded
// features (f >= F) read the zero slot z~[D+1].
// ------------------------------------------------------------------------------------------
template <int D, int F>
struct FeatAB {
  static constexpr int find_a() {
    int a = 0, base = 0;
    while (a <= D && base + (D + 1 - a) <= F) { base += D + 1 - a; ++a; }
    return a;
  }
  static constexpr int base_of(int a) {
    int b = 0;
    for (int i = 0; i < a; ++i) b += D + 1 - i;
    return b;
  }
  static constexpr bool pad = F >= (D + 1) * (D + 2) / 2;
  static constexpr int a = pad ? D + 1 : find_a();
  static constexpr int b = pad ? D + 1 : a + (F - base_of(a));
};

// lane (row = lane & 31, parity = lane >> 5) of wave W writes features W*FW + 2i + parity, i < FW/2
template <int D, int FW, int W, int... I>
__device__ __forceinline__ void build_features_static(const double (&z)[D + 2], double* __restrict__ prow,
                                                      const bool parity, std::integer_sequence<int, I...>) {
  ((prow[W * FW + 2 * I] = (parity ? z[FeatAB<D, W * FW + 2 * I + 1>::a] : z[FeatAB<D, W * FW + 2 * I>::a]) *
                           (parity ? z[FeatAB<D, W * FW + 2 * I + 1>::b] : z[FeatAB<D, W * FW + 2 * I>::b])),
   ...);
}

// ------------------------------------------------------------------------------------------
// Per-datum normalisation over k of one 32-row tile held in LDS as Lt[row][component]:
// 8 lanes per datum, 2*K16 consecutive components per lane (<= 8 here: RBW = 1), fully unrolled.
// Softmax -> r written back in place, or inverse-CDF categorical draw -> label (LDS + HBM).
// ------------------------------------------------------------------------------------------
template <int RBW, int MODE, bool E2K = false>
__device__ __forceinline__ void normalise_tile(const KernelArgs& a, double* __restrict__ Lt, const int LS,
                                               const double* __restrict__ etab, const int K, const int K16,
                                               const int64_t N, const int64_t n0, const int wave, const int lane,
                                               const bool gibbs, double* const out_logp, double* const out_resp,
                                               double* const out_lse, double& sc_lse, double& sc_rl, double& sc_prod,
                                               int* __restrict__ labs, PhiloxBatch& pb,
                                               const int64_t tstride) {
        static_assert(RBW == 1, "register variant: at most 8 components per lane");
        const int pt = 8 * wave + (lane & 7), part = lane >> 3;
        const int CPP = 2 * K16, k0 = part * CPP;
        const int64_t n = n0 + pt;
        const bool valid = n < N;
        double* row = Lt + pt * LS + k0;

        auto body = [&](auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        // Padding components (k >= K inside the last row block) carry l = kPadLogDensity from the operand
        // image, so only the HBM table writes test k < K.  FULL: this lane owns all 8 slots (K16 = 4) and no
        // slot test is compiled at all; otherwise slots c >= CPP belong to the next lane and are masked.
        auto ok = [&](int c) { return FULL || c < CPP; };
        // x[] holds l, then exp(l - max), then the weight written back — one register array
        double x[8], lsave[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] = ok(c) ? row[c] : -INFINITY;
        if (out_logp && valid) {
#pragma unroll
          for (int c = 0; c < 8; ++c)
            if (c < CPP && k0 + c < K) out_logp[(int64_t)(k0 + c) * N + n] = x[c];
        }
        double m = tree_max8(x);
        m = fmax(m, __shfl_xor(m, 8));
        m = fmax(m, __shfl_xor(m, 16));
        m = fmax(m, __shfl_xor(m, 32));

#pragma unroll
        for (int c = 0; c < 8; ++c) {
          // (a component switched off by c_k = -inf carries l = -1e300: its e is exp(-707) = 8e-308, not 0, and
          // e * l would add -8e-8 per datum to sum_k r l — its term is 0 * (-inf) := 0, like the padding's)
          if constexpr (MODE == kGeneric) lsave[c] = (c < CPP && k0 + c < K && x[c] > kOffLogDensity) ? x[c] : 0.0;
          x[c] = E2K ? exp_nonpos_t2048c(x[c] - m, etab) : exp_nonpos(x[c] - m, etab);      // masked / padding slots -> 0
        }
        double sel = 0.0;
        if constexpr (MODE == kGeneric) {   // sum_k e l only feeds the entropy split of the ELBO (scalars[1..2])
          double t[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) t[c] = x[c] * lsave[c];
          sel = tree_sum8(t);
          sel += __shfl_xor(sel, 8);
          sel += __shfl_xor(sel, 16);
          sel += __shfl_xor(sel, 32);
        }

        if (!gibbs) {
          double ssum = tree_sum8(x);
          ssum += __shfl_xor(ssum, 8);
          ssum += __shfl_xor(ssum, 16);
          ssum += __shfl_xor(ssum, 32);
          // 1 / sum: v_rcp_f64 seed + two Newton steps (5 dependent f64 ops instead of the IEEE divide's
          // ~12; every one of them waits for a matrix-pipe slot); relative error <= 1 ulp-ish (2^-52).
          double inv = __builtin_amdgcn_rcp(ssum);
          inv = fma(fma(-ssum, inv, 1.0), inv, inv);
          inv = fma(fma(-ssum, inv, 1.0), inv, inv);
          if constexpr (MODE == kGeneric) {
            const double lse = m + log(ssum);
            if (part == 0 && valid) {
              sc_lse += lse;
              sc_rl += sel * inv;
              if (out_lse) out_lse[n] = lse;
            }
          } else {
            // fast modes only need sum_n lse_n = sum_n m_n + log prod_n ssum_n: the product of the per-datum
            // sums (each in [1, K]) is accumulated and its log taken once per 64 tiles by the caller.
            if (part == 0 && valid) {
              sc_lse += m;
              sc_prod *= ssum;
            }
          }
          const double scale = valid ? inv : 0.0;
          // per-row weights (generic mode only; a.u doubles as the weight vector of a mean-field pass): the
          // statistics are those of r_kn w_n, the tables and the ELBO scalars those of r_kn
          // (mimo/mixtures/hgmm.py:199-207: resp * weights feeds the update, the unweighted resp the bound)
          double wrow = 1.0;
          if constexpr (MODE == kGeneric) wrow = (a.u && valid) ? a.u[n] : 1.0;
          if (out_resp && valid) {
#pragma unroll
            for (int c = 0; c < 8; ++c)
              if (c < CPP && k0 + c < K) out_resp[(int64_t)(k0 + c) * N + n] = x[c] * scale;
          }
          const double wscale = MODE == kGeneric ? scale * wrow : scale;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            x[c] *= wscale;
            if (ok(c)) row[c] = x[c];
          }
        } else {
          // inverse-CDF draw (mimo/utils/stats.py:10-17): label = #{k : u * cum[K-1] > cum[k]} with
          // cum = cumsum_k exp(l_k - lse).  The count is invariant to the common factor 1/sum e, so the
          // UNNORMALISED cumulative sums E_k = sum_{j<=k} e_j are compared with u * E_K (same labels up
          // to last-bit ties; no per-component scaling, no one-hot table: only the label is kept).
          scan8(x);                                  // local inclusive cumulative sums
          const double cum = x[7];
          double incl = cum;  // inclusive scan over the 8 parts of this datum
          {
            double v = __shfl_up(incl, 8);  if (part >= 1) incl += v;
            v = __shfl_up(incl, 16);        if (part >= 2) incl += v;
            v = __shfl_up(incl, 32);        if (part >= 4) incl += v;
          }
          double excl = __shfl_up(incl, 8);
          if (part == 0) excl = 0.0;
          const double ctot = __shfl(excl + cum, 56 + (lane & 7));  // == last cumulative value
          if constexpr (MODE == kGeneric) {
            const double lse = m + log(ctot);
            if (part == 0 && valid) {
              sc_lse += lse;
              sc_rl += sel / ctot;
              if (out_lse) out_lse[n] = lse;
            }
          } else {
            if (part == 0 && valid) {
              sc_lse += m;
              sc_prod *= ctot;
            }
          }
          const double uu = a.u ? (valid ? a.u[n] : 0.0) : philox_for_tile(pb, a, n, lane, tstride);
          const double tl = uu * ctot - excl;   // threshold in this lane's local cumulative scale
          int cnt = 0;
#pragma unroll
          for (int c = 0; c < 8; ++c)     // (padding slots can only be counted above the last real one: capped below)
            cnt += (ok(c) && tl > x[c]) ? 1 : 0;
          cnt += __shfl_xor(cnt, 8);
          cnt += __shfl_xor(cnt, 16);
          cnt += __shfl_xor(cnt, 32);
          const int label = cnt < K ? cnt : K - 1;
          if (part == 0) {
            labs[pt] = valid ? label : -1;      // the statistics phase builds its one-hot operand from this
            if (valid && a.labels) a.labels[n] = label;
          }
        }
        };
        if (CPP == 8) body(std::true_type{});
        else body(std::false_type{});
}

// Same contract for lanes that own up to 8*RBW components (RBW > 1), processed in chunks of 8.
//   pass 1: max (8 independent chains across the chunks, then a tree)
//   pass 2: e = exp(l - max) written back in place, chunk totals T[ch] (trees) kept in registers
//   pass 3: softmax: e scaled by 1/sum.  Gibbs: the chunk that holds the crossing is located from the
//           chunk totals, and only ITS eight e are read back, scanned and compared — 8 LDS reads and 12
//           compares instead of 32 + 32.
template <int RBW, int MODE>
__device__ __forceinline__ void normalise_tile_chunked(const KernelArgs& a, double* __restrict__ Lt, const int LS,
                                                       const double* __restrict__ etab, const int K, const int K16,
                                                       const int64_t N, const int64_t n0, const int wave,
                                                       const int lane, const bool gibbs, double* const out_logp,
                                                       double* const out_resp, double* const out_lse,
                                                       double& sc_lse, double& sc_rl, double& sc_prod,
                                                       int* __restrict__ labs, PhiloxBatch& pb,
                                                       const int64_t tstride) {
  const int pt = 8 * wave + (lane & 7), part = lane >> 3;
  const int CPP = 2 * K16, k0 = part * CPP;
  const int64_t n = n0 + pt;
  const bool valid = n < N;
  double* row = Lt + pt * LS + k0;
  // padding components carry l = kPadLogDensity (operand image): only HBM table writes test k < K.
  // FULL: the lane owns all 8*RBW slots (K16 = 4*RBW) and no slot test is compiled.
  auto body = [&](auto full_c) {
  constexpr bool FULL = decltype(full_c)::value;
  auto ok = [&](int c) { return FULL || c < CPP; };
  auto active = [&](int c) { return ok(c) && k0 + c < K; };

  double mv[8];
#pragma unroll
  for (int cc = 0; cc < 8; ++cc) mv[cc] = -INFINITY;
#pragma unroll
  for (int ch = 0; ch < RBW; ++ch) {
    if (FULL || 8 * ch < CPP) {
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) {
        const int c = 8 * ch + cc;
        const double l = ok(c) ? row[c] : -INFINITY;
        if (out_logp && valid && active(c)) out_logp[(int64_t)(k0 + c) * N + n] = l;
        mv[cc] = fmax(mv[cc], l);
      }
    }
  }
  double m = tree_max8(mv);
  m = fmax(m, __shfl_xor(m, 8));
  m = fmax(m, __shfl_xor(m, 16));
  m = fmax(m, __shfl_xor(m, 32));

  double T[RBW], selv[RBW];
#pragma unroll
  for (int ch = 0; ch < RBW; ++ch) {
    T[ch] = 0.0;
    selv[ch] = 0.0;
    if (FULL || 8 * ch < CPP) {
      double x[8], lc[8];
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) lc[cc] = ok(8 * ch + cc) ? row[8 * ch + cc] : -INFINITY;
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) x[cc] = exp_nonpos(lc[cc] - m, etab);
      T[ch] = tree_sum8(x);
      if constexpr (MODE == kGeneric) {
        double t[8];
#pragma unroll
        for (int cc = 0; cc < 8; ++cc)     // (switched-off components: see normalise_tile)
          t[cc] = x[cc] * ((active(8 * ch + cc) && lc[cc] > kOffLogDensity) ? lc[cc] : 0.0);
        selv[ch] = tree_sum8(t);
      }
#pragma unroll
      for (int cc = 0; cc < 8; ++cc)        // e replaces l in place (softmax: scaled in pass 3; Gibbs: the
        if (ok(8 * ch + cc)) row[8 * ch + cc] = x[cc];   // crossing chunk is read back in pass 3)
    }
    __builtin_amdgcn_sched_barrier(0);   // one chunk of exp chains in flight at a time (register pressure)
  }
  // exclusive prefix of the chunk totals (base[ch] = sum of the chunks before ch) and the lane total
  double base[RBW + 1];
  base[0] = 0.0;
#pragma unroll
  for (int ch = 0; ch < RBW; ++ch) base[ch + 1] = base[ch] + T[ch];
  const double cum = base[RBW];
  double sel = 0.0;
  if constexpr (MODE == kGeneric) {
#pragma unroll
    for (int ch = 0; ch < RBW; ++ch) sel += selv[ch];
    sel += __shfl_xor(sel, 8);
    sel += __shfl_xor(sel, 16);
    sel += __shfl_xor(sel, 32);
  }

  if (!gibbs) {
    double ssum = cum;
    ssum += __shfl_xor(ssum, 8);
    ssum += __shfl_xor(ssum, 16);
    ssum += __shfl_xor(ssum, 32);
    double inv = __builtin_amdgcn_rcp(ssum);
    inv = fma(fma(-ssum, inv, 1.0), inv, inv);
    inv = fma(fma(-ssum, inv, 1.0), inv, inv);
    if constexpr (MODE == kGeneric) {
      const double lse = m + log(ssum);
      if (part == 0 && valid) {
        sc_lse += lse;
        sc_rl += sel * inv;
        if (out_lse) out_lse[n] = lse;
      }
    } else {
      if (part == 0 && valid) {
        sc_lse += m;
        sc_prod *= ssum;
      }
    }
    const double scale = valid ? inv : 0.0;
    double wrow = 1.0;   // per-row weights: see normalise_tile
    if constexpr (MODE == kGeneric) wrow = (a.u && valid) ? a.u[n] : 1.0;
#pragma unroll
    for (int c = 0; c < 8 * RBW; ++c) {
      if (ok(c)) {
        const double r = row[c] * scale;
        row[c] = MODE == kGeneric ? r * wrow : r;
        if (out_resp && valid && k0 + c < K) out_resp[(int64_t)(k0 + c) * N + n] = r;
      }
      if ((c & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    // scale-invariant inverse CDF on the unnormalised cumulative sums (see normalise_tile)
    double incl = cum;
    {
      double v = __shfl_up(incl, 8);  if (part >= 1) incl += v;
      v = __shfl_up(incl, 16);        if (part >= 2) incl += v;
      v = __shfl_up(incl, 32);        if (part >= 4) incl += v;
    }
    double excl = __shfl_up(incl, 8);
    if (part == 0) excl = 0.0;
    const double ctot = __shfl(excl + cum, 56 + (lane & 7));
    if constexpr (MODE == kGeneric) {
      const double lse = m + log(ctot);
      if (part == 0 && valid) {
        sc_lse += lse;
        sc_rl += sel / ctot;
        if (out_lse) out_lse[n] = lse;
      }
    } else {
      if (part == 0 && valid) {
        sc_lse += m;
        sc_prod *= ctot;
      }
    }
    const double uu = a.u ? (valid ? a.u[n] : 0.0) : philox_uniform(a.seed, (uint64_t)(a.row0 + n), a.sweep);
    const double tl = uu * ctot - excl;
    // chunk of the crossing: j = #{ch : tl > cumulative sum at the END of chunk ch}
    int j = 0;
#pragma unroll
    for (int ch = 0; ch < RBW; ++ch) j += (tl > base[ch + 1]) ? 1 : 0;
    // components of this lane that exist at all (the count if tl lies above every cumulative sum)
    int cnt;
    if ((!FULL && 8 * j >= CPP) || j >= RBW) {
      cnt = CPP;      // above every cumulative sum of this lane (padding slots included: capped below)
    } else {
      double bj = base[0];
#pragma unroll
      for (int ch = 1; ch < RBW; ++ch) bj = (j == ch) ? base[ch] : bj;
      const double* rj = row + 8 * j;
      double x[8];
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) x[cc] = (FULL || 8 * j + cc < CPP) ? rj[cc] : 0.0;
      scan8(x);
      const double tj = tl - bj;
      cnt = 8 * j;
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) cnt += ((FULL || 8 * j + cc < CPP) && tj > x[cc]) ? 1 : 0;
    }
    cnt += __shfl_xor(cnt, 8);
    cnt += __shfl_xor(cnt, 16);
    cnt += __shfl_xor(cnt, 32);
    const int label = cnt < K ? cnt : K - 1;
    if (part == 0) {
      labs[pt] = valid ? label : -1;
      if (valid && a.labels) a.labels[n] = label;
    }
  }
  };
  if (CPP == 8 * RBW) body(std::true_type{});
  else body(std::false_type{});
}

}  // namespace mimo
