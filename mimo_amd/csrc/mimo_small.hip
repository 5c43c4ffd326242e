// gfx950 kernel for the SMALL shapes of the same path: Dz <= 4 (F <= 15 features) and K <= 32 — the shapes of the
// reference's own examples (examples/gmm/toy: D = 2, K = 4; examples/ilr/evaluate_sine.py: dx = dy = 1;
// examples/gmm/sine: K = 25).  Through the 16-padded float64 MFMA tiles of mimo_kernels.hip such a sweep costs what
// a K = 16, F = 16 sweep costs (0.9 ms for 1e7 rows at D = 2, K = 4: 1.3 % of the HBM roof these shapes are bound
// by: 16 bytes per row).  Here the work is plain float64 VALU, one datum per lane:
//
//   lane = (row rl = lane / G, component group g = lane % G); the lane owns KL components k = g KL .. g KL + KL - 1
//   Theta rows of its components and the KL x F statistic accumulators live in registers for the whole kernel
//   per row:  features (Dz(Dz+1)/2 products)  ->  l_k = Theta_k . phi  (F - 1 fma per component)
//             max / sum over the row's G lanes (xor shuffles)  ->  softmax or inverse-CDF label draw
//             S_k += r_k phi  (F - 1 fma + 1 add per component)
//   rows stream through a persistent grid, U passes of 64 / G rows per wave in flight (next chunk prefetched)
//
// D = 2, K = 4: ~135 float64 instructions and 16 bytes per row — at the 4-cycle issue rate of the f64 VALU the
// instruction stream and the HBM stream need about the same time; DESIGN.md section 4 has the measured numbers.
// The accumulators are reduced per workgroup through LDS in a FIXED order and written as one partial block in the
// layout of the tile kernels ([16 K16][16] + 4 scalars), so reduce_partials / unpack_stats finish the job and the
// results are run-to-run bit-identical.
//
// Reference behaviour reproduced: see the header of mimo_kernels.hip (same tables, same draw, same statistics).
#include "mimo_device.h"

#include <type_traits>

namespace mimo {

namespace {

template <int DZ>
__device__ __forceinline__ void load_row(const double* __restrict__ p, double (&z)[DZ]) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  if constexpr (DZ == 1) {
    z[0] = p[0];
  } else if constexpr (DZ == 2) {
    const d2 v = *reinterpret_cast<const d2*>(p);          // 16-byte rows: the host checks the base alignment
    z[0] = v.x; z[1] = v.y;
  } else if constexpr (DZ == 3) {
    z[0] = p[0]; z[1] = p[1]; z[2] = p[2];
  } else {
    const d2 v = *reinterpret_cast<const d2*>(p), w = *reinterpret_cast<const d2*>(p + 2);
    z[0] = v.x; z[1] = v.y; z[2] = w.x; z[3] = w.y;
  }
}

// feature f of the pair (a, b), a <= b <= DZ over z~ = [z, 1]: row-major upper triangle (= feat_index of mimo_kernels.h)
constexpr int sfeat(int DZ, int a, int b) { return a * (DZ + 1) - a * (a - 1) / 2 + (b - a); }

}  // namespace

#ifndef MIMO_SMALL_U1
#define MIMO_SMALL_U1 2      // chunks of U passes in flight per lane, G = 1 and Dz <= 2
#endif
#ifndef MIMO_SMALL_OCC1
#define MIMO_SMALL_OCC1 3    // workgroups per CU the G = 1, Dz <= 2 kernels are compiled for
#endif
template <int DZ, int KL, int G, int MODE>
__global__ __launch_bounds__(kWG, (G == 1 && DZ <= 2) ? MIMO_SMALL_OCC1 : (DZ == 1 ? 3 : 2))
void small_kernel(const KernelArgs a) {
  constexpr int F = (DZ + 1) * (DZ + 2) / 2;
  constexpr int RPW = 64 / G;               // rows per wave pass
  constexpr int U = (DZ <= 2 && G == 1) ? MIMO_SMALL_U1 : 2;   // passes per chunk (loads in flight per lane)
  constexpr int CH = U * RPW;               // rows per wave chunk
  constexpr int NV = KL * F, VB = 16;
  constexpr bool kEstep = MODE <= kGeneric;
  static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16, "G divides 16");

#ifndef MIMO_SMALL_EXP2048
#define MIMO_SMALL_EXP2048 1      // 2048-entry exp table (13 instead of 15 instructions per exponential, 16 KB of LDS)
#endif
#if MIMO_SMALL_EXP2048
  __shared__ double etab[kExpTab];
#else
  __shared__ double etab[64];
#endif
  __shared__ double red[VB][kWG + 1];
  __shared__ double part[VB][16];
  __shared__ double sred[8];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = G == 1 ? 0 : lane % G, rl = lane / G;
  const int K = a.K;
  const int64_t N = a.N;
  const bool gibbs = MODE == kFastVI ? false : MODE == kFastGibbs ? true : a.gibbs != 0;
  const bool do_stats = (MODE == kFastVI || MODE == kFastGibbs || MODE > kGeneric) ? true : a.do_stats != 0;
  double* const out_logp = MODE == kGeneric ? a.logp : nullptr;
  double* const out_resp = MODE == kGeneric ? a.resp : nullptr;
  double* const out_lse = MODE == kGeneric ? a.lse : nullptr;
#if MIMO_SMALL_EXP2048
  for (int e = tid; e < kExpTab; e += kWG) etab[e] = exp_tab_entry_c(e);
#else
  if (tid < 64) etab[tid] = exp2((double)tid * (1.0 / 64.0));
#endif

  // Theta rows of this lane's components ([G KL][F] row-major, padding components carry c = -1e300): G = 1 makes
  // the addresses wave-uniform and the rows live in SGPRs
  double th[KL][F];
  if constexpr (kEstep) {
    // G = 1: the KL F <= 40 numbers travel in the kernel arguments (no staging copy, no H2D transfer in front of the launch)
    const double* thp = (G == 1 && KL * F <= kThetaInline) ? a.theta_inline : a.theta;
#pragma unroll
    for (int c = 0; c < KL; ++c)
#pragma unroll
      for (int f = 0; f < F; ++f) th[c][f] = thp[(size_t)(g * KL + c) * F + f];
  }
  double acc[KL][F];
#pragma unroll
  for (int c = 0; c < KL; ++c)
#pragma unroll
    for (int f = 0; f < F; ++f) acc[c][f] = 0.0;

  double sc_lse = 0.0, sc_rl = 0.0, sc_prod = 1.0;
  wg_sync();

  const int64_t nwaves = (int64_t)gridDim.x * 4, wv = (int64_t)blockIdx.x * 4 + wave;
  const int64_t nchunks = (N + CH - 1) / CH;

  const int64_t nfull = N / CH;      // chunks whose rows all exist
  auto load_chunk = [&](int64_t chunk, double (&z)[U][DZ]) {
#if defined(MIMO_SMALL_EXPERIMENT) && MIMO_SMALL_EXPERIMENT == 1      // what-if: no memory traffic
    for (int u = 0; u < U; ++u) for (int d = 0; d < DZ; ++d) z[u][d] = 1e-3 * (double)(lane + u + d) + 1e-9 * (double)chunk;
    return;
#endif
    if (chunk < nfull) {             // wave-uniform
#pragma unroll
      for (int u = 0; u < U; ++u) load_row<DZ>(a.Z + (chunk * CH + u * RPW + rl) * DZ, z[u]);
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int64_t n = chunk * CH + u * RPW + rl;
        n = n < N ? n : N - 1;                // rows past N re-read the last row; their weight is zeroed below
        load_row<DZ>(a.Z + n * DZ, z[u]);
      }
    }
  };

  // CHECK = false: every row of the chunk exists (all chunks but the last one of the data set): no validity selects
  auto process = [&](auto check_c, const int64_t n, const double (&z)[DZ]) {
    constexpr bool CHECK = decltype(check_c)::value;
#if defined(MIMO_SMALL_EXPERIMENT) && MIMO_SMALL_EXPERIMENT == 2      // what-if: memory traffic only
    for (int d = 0; d < DZ; ++d) acc[0][d] += z[d];
    return;
#endif
    const bool valid = CHECK ? n < N : true;
    const int64_t nn = (CHECK && !valid) ? N - 1 : n;
    // features z_a z_b (a <= b < DZ); (a, DZ) is z_a itself and (DZ, DZ) the constant 1
    double phi[F];
#pragma unroll
    for (int i = 0; i < DZ; ++i) {
#pragma unroll
      for (int j = i; j < DZ; ++j) phi[sfeat(DZ, i, j)] = z[i] * z[j];
      phi[sfeat(DZ, i, DZ)] = z[i];
    }
    phi[F - 1] = 1.0;

    double r[KL];
    if constexpr (MODE == kModeWeights) {
#pragma unroll
      for (int c = 0; c < KL; ++c) {
        const int k = g * KL + c;
        r[c] = (valid && k < K) ? a.resp[(int64_t)k * N + nn] : 0.0;
      }
    } else if constexpr (MODE == kModeLabels) {
      const int lab = valid ? a.labels[nn] : -1;
#pragma unroll
      for (int c = 0; c < KL; ++c) r[c] = lab == g * KL + c ? 1.0 : 0.0;
    } else {
      double l[KL];
#pragma unroll
      for (int c = 0; c < KL; ++c) {
        double v = th[c][F - 1];
#pragma unroll
        for (int f = 0; f < F - 1; ++f) v = fma(th[c][f], phi[f], v);
        l[c] = v;
      }
      if (out_logp && valid) {
#pragma unroll
        for (int c = 0; c < KL; ++c)
          if (g * KL + c < K) out_logp[(int64_t)(g * KL + c) * N + n] = l[c];
      }
      double m = l[0];
#pragma unroll
      for (int c = 1; c < KL; ++c) m = fmax(m, l[c]);
#pragma unroll
      for (int s = 1; s < G; s <<= 1) m = fmax(m, __shfl_xor(m, s));
      double e[KL];
#pragma unroll
#if MIMO_SMALL_EXP2048
      for (int c = 0; c < KL; ++c) e[c] = exp_nonpos_t2048c(l[c] - m, etab);
#else
      for (int c = 0; c < KL; ++c) e[c] = exp_nonpos(l[c] - m, etab);
#endif
      double sel = 0.0;
      if constexpr (MODE == kGeneric) {     // sum_k e l feeds the entropy split of the ELBO scalars (switched-off / padding: 0)
#pragma unroll
        for (int c = 0; c < KL; ++c) sel += e[c] * ((g * KL + c < K && l[c] > kOffLogDensity) ? l[c] : 0.0);
#pragma unroll
        for (int s = 1; s < G; s <<= 1) sel += __shfl_xor(sel, s);
      }
      if (!gibbs) {
        double ssum = e[0];
#pragma unroll
        for (int c = 1; c < KL; ++c) ssum += e[c];
#pragma unroll
        for (int s = 1; s < G; s <<= 1) ssum += __shfl_xor(ssum, s);
        double inv = __builtin_amdgcn_rcp(ssum);
        inv = fma(fma(-ssum, inv, 1.0), inv, inv);
        inv = fma(fma(-ssum, inv, 1.0), inv, inv);
        if constexpr (MODE == kGeneric) {
          const double lse = m + log(ssum);
          if (g == 0 && valid) {
            sc_lse += lse;
            sc_rl += sel * inv;
            if (out_lse) out_lse[n] = lse;
          }
        } else {
          if (g == 0 && valid) { sc_lse += m; sc_prod *= ssum; }
        }
        const double scale = valid ? inv : 0.0;
        double wrow = 1.0;      // per-row weights (a.u of a mean-field pass): statistics of r w, tables and scalars of r
        if constexpr (MODE == kGeneric) wrow = (a.u && valid) ? a.u[nn] : 1.0;
#pragma unroll
        for (int c = 0; c < KL; ++c) {
          const double rr = e[c] * scale;
          if (out_resp && valid && g * KL + c < K) out_resp[(int64_t)(g * KL + c) * N + n] = rr;
          r[c] = MODE == kGeneric ? rr * wrow : rr;
        }
      } else {
        // inverse-CDF draw on the unnormalised cumulative sums (mimo/utils/stats.py:10-17; see normalise_tile)
        double E[KL];
        E[0] = e[0];
#pragma unroll
        for (int c = 1; c < KL; ++c) E[c] = E[c - 1] + e[c];
        const double cum = E[KL - 1];
        double incl = cum;      // inclusive scan over the G lanes of this row (they are adjacent: lane = rl G + g)
#pragma unroll
        for (int s = 1; s < G; s <<= 1) {
          const double v = __shfl_up(incl, s);
          if (g >= s) incl += v;
        }
        double excl = G > 1 ? __shfl_up(incl, 1) : 0.0;
        if (g == 0) excl = 0.0;
        const double ctot = G > 1 ? __shfl(incl, (lane / G) * G + (G - 1)) : incl;
        if constexpr (MODE == kGeneric) {
          const double lse = m + log(ctot);
          if (g == 0 && valid) {
            sc_lse += lse;
            sc_rl += sel / ctot;
            if (out_lse) out_lse[n] = lse;
          }
        } else {
          if (g == 0 && valid) { sc_lse += m; sc_prod *= ctot; }
        }
        const double uu = a.u ? a.u[nn] : philox_uniform(a.seed, (uint64_t)(a.row0 + nn), a.sweep);
        const double tl = uu * ctot - excl;
        int cnt = 0;
#pragma unroll
        for (int c = 0; c < KL; ++c) cnt += tl > E[c] ? 1 : 0;
#pragma unroll
        for (int s = 1; s < G; s <<= 1) cnt += __shfl_xor(cnt, s);
        const int label = cnt < K ? cnt : K - 1;
        if (g == 0 && valid && a.labels) a.labels[n] = label;
#pragma unroll
        for (int c = 0; c < KL; ++c) r[c] = (valid && label == g * KL + c) ? 1.0 : 0.0;
      }
    }
    if (do_stats) {
#pragma unroll
      for (int c = 0; c < KL; ++c) {
#pragma unroll
        for (int f = 0; f < F - 1; ++f) acc[c][f] = fma(r[c], phi[f], acc[c][f]);
        acc[c][F - 1] += r[c];
      }
    }
  };

  {
    // two chunks per iteration, ping-pong between two register sets (no copies): while chunk i is processed the
    // loads of chunk i + nwaves are in flight
    double za[U][DZ], zb[U][DZ];
    auto run_chunk = [&](int64_t chunk, const double (&z)[U][DZ]) {
      if (chunk < nfull) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          process(std::false_type{}, chunk * CH + u * RPW + rl, z[u]);
          __builtin_amdgcn_sched_barrier(0);     // one row at a time: interleaved rows multiply the live temporaries
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          process(std::true_type{}, chunk * CH + u * RPW + rl, z[u]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    // sum_n lse_n = sum_n m_n + log prod_n ssum_n in the fast modes: every factor lies in [1, G KL], so the product
    // of up to 1000 / log2(G KL) rows stays inside the float64 range; it is folded into the sum with ONE log per that
    // many rows, at loop level (a log inside the row code costs its constants in registers, 8 copies of it)
    constexpr int kBits = G * KL <= 4 ? 2 : G * KL <= 8 ? 3 : G * KL <= 16 ? 4 : 5;
    constexpr int kFlushIters = 1000 / kBits / (2 * U);
    int since_flush = 0;
    int64_t chunk = wv;
    if (chunk < nchunks) load_chunk(chunk, za);
    while (chunk < nchunks) {
      if (chunk + nwaves < nchunks) load_chunk(chunk + nwaves, zb);
      run_chunk(chunk, za);
      chunk += nwaves;
      if (chunk < nchunks) {
        if (chunk + nwaves < nchunks) load_chunk(chunk + nwaves, za);
        run_chunk(chunk, zb);
        chunk += nwaves;
      }
      if constexpr (MODE == kFastVI || MODE == kFastGibbs) {
        if (++since_flush == kFlushIters) {
          sc_lse += log(sc_prod);
          sc_prod = 1.0;
          since_flush = 0;
        }
      }
    }
  }

  // ---- per-workgroup partial block: fixed-order reduction through LDS, VB accumulators at a time -------------
  const size_t pstride = (size_t)a.K16 * 16 * a.F16_total + 4;
  double* P = a.partials + (size_t)blockIdx.x * pstride;
#pragma unroll
  for (int v0 = 0; v0 < NV; v0 += VB) {
    wg_sync();
#pragma unroll
    for (int i = 0; i < VB; ++i)
      if (v0 + i < NV) red[i][tid] = acc[(v0 + i) / F][(v0 + i) % F];
    wg_sync();
    {   // thread (value i, part p): lanes p, p + 16, ... of the workgroup — all of component group p % G
      const int i = tid >> 4, p = tid & 15;
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j) s += red[i][p + 16 * j];
      part[i][p] = s;
    }
    wg_sync();
    if (tid < VB * G) {
      const int i = tid / G, gg = tid % G;
      if (v0 + i < NV) {
        double s = 0.0;
        for (int p = gg; p < 16; p += G) s += part[i][p];
        const int v = v0 + i, c = v / F, f = v - c * F;
        P[(size_t)(gg * KL + c) * a.F16_total + f] = s;
      }
    }
  }
  if constexpr (MODE == kFastVI || MODE == kFastGibbs) sc_lse += log(sc_prod);
  sc_lse = wave_sum(sc_lse);
  sc_rl = wave_sum(sc_rl);
  if (lane == 0) { sred[2 * wave] = sc_lse; sred[2 * wave + 1] = sc_rl; }
  wg_sync();
  if (tid == 0 && a.write_scalars) {
    double* Ps = P + (size_t)a.K16 * 16 * a.F16_total;
    Ps[0] = (sred[0] + sred[2]) + (sred[4] + sred[6]);
    Ps[1] = (sred[1] + sred[3]) + (sred[5] + sred[7]);
    Ps[2] = MODE == kGeneric ? 1.0 : 0.0;
    Ps[3] = 0.0;
  }
}

// ------------------------------------------------------------------------------------------
// Categorical draw from a caller-supplied (K, N) table of log-probabilities, one thread per column
// (mimo/utils/stats.py:8-21 with axis = 0): label = #{k : u cum_K > cum_k}, cum = cumsum_k exp(l - max), and
// optionally lognorm_n = logsumexp_k l.  Bound by the one-and-a-half reads of the table (pass 1: max and total,
// pass 2: cumulative sums against the threshold); K-major rows make every read of a wave contiguous.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWG) void sample_table_kernel(const double* __restrict__ logp, int K, int64_t N,
                                                           const double* __restrict__ u, uint64_t seed, uint64_t sweep,
                                                           int64_t row0, int32_t* __restrict__ labels,
                                                           double* __restrict__ lognorms) {
  __shared__ double etab[64];
  if (threadIdx.x < 64) etab[threadIdx.x] = exp2((double)threadIdx.x * (1.0 / 64.0));
  wg_sync();
  const int64_t n = (int64_t)blockIdx.x * kWG + threadIdx.x;
  if (n >= N) return;
  double m = logp[n];
  for (int k = 1; k < K; ++k) m = fmax(m, logp[(int64_t)k * N + n]);
  double tot = 0.0;
  for (int k = 0; k < K; ++k) tot += exp_nonpos(logp[(int64_t)k * N + n] - m, etab);
  const double uu = u ? u[n] : philox_uniform(seed, (uint64_t)(row0 + n), sweep);
  const double thr = uu * tot;
  double cum = 0.0;
  int cnt = 0;
  for (int k = 0; k < K; ++k) {
    cum += exp_nonpos(logp[(int64_t)k * N + n] - m, etab);
    cnt += thr > cum ? 1 : 0;
  }
  labels[n] = cnt < K ? cnt : K - 1;
  if (lognorms) lognorms[n] = m + log(tot);
}

hipError_t launch_sample_table(const double* logp, int K, int64_t N, const double* u, uint64_t seed, uint64_t sweep,
                               int64_t row0, int32_t* labels, double* lognorms, hipStream_t stream) {
  if (N <= 0) return hipSuccess;
  hipLaunchKernelGGL(sample_table_kernel, dim3((unsigned)((N + kWG - 1) / kWG)), dim3(kWG), 0, stream, logp, K, N, u, seed,
                     sweep, row0, labels, lognorms);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Random initial responsibilities on the device (the drivers' randomize=True start, mimo/mixtures/gmm.py:265-267:
// resp = rand(K, N); resp /= resp.sum(0)) without drawing K N uniforms on the host and shipping them over PCIe:
// r[k, n] = v_kn / sum_j v_jn with v_kn the Philox4x32-10 uniform of key `seed`, counter (row0 + n, k) — a stated
// stream of its own (numpy's generator cannot be continued on the device), independent of the number of shards.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWG) void random_resp_kernel(double* __restrict__ resp, int K, int64_t N, uint64_t seed,
                                                          int64_t row0) {
  const int64_t n = (int64_t)blockIdx.x * kWG + threadIdx.x;
  if (n >= N) return;
  double tot = 0.0;
  for (int k = 0; k < K; ++k) {
    // (0, 1]: a column of zeros cannot happen
    const double v = philox_uniform(seed, (uint64_t)(row0 + n), (uint64_t)k) + 1.1102230246251565e-16;
    resp[(int64_t)k * N + n] = v;
    tot += v;
  }
  const double inv = 1.0 / tot;
  for (int k = 0; k < K; ++k) resp[(int64_t)k * N + n] *= inv;
}

hipError_t launch_random_resp(double* resp, int K, int64_t N, uint64_t seed, int64_t row0, hipStream_t stream) {
  if (N <= 0) return hipSuccess;
  hipLaunchKernelGGL(random_resp_kernel, dim3((unsigned)((N + kWG - 1) / kWG)), dim3(kWG), 0, stream, resp, K, N, seed, row0);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Rows with missing values.  The reference drops rows that hold a NaN from every statistic
// (mimo/distributions/gaussian.py:493-494: idx = ~isnan(data).any(axis=1)) and gives them the normaliser-only
// log-density (gaussian.py:512-520: nan_to_num, then the data-dependent part of the row is set to 0) — which is the
// canonical form at z = 0.  nan_scan zeroes such rows in the library's OWN copy of the data and writes the row mask
// (1 = complete row) that the passes then use as per-row weights of the statistics; the helpers below apply the mask to
// label vectors and weight tables.  NaN is tested on the bit pattern (this file is built with -fno-honor-nans).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool is_nan_bits(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v) & 0x7fffffffffffffffull;
  return b > 0x7ff0000000000000ull;
}

__global__ __launch_bounds__(kWG) void nan_scan_kernel(double* __restrict__ Z, int64_t N, int D, double* __restrict__ mask,
                                                       unsigned long long* __restrict__ count, int write) {
  const int64_t n = (int64_t)blockIdx.x * kWG + threadIdx.x;
  if (n >= N) return;
  bool bad = false;
  for (int d = 0; d < D; ++d) bad = bad || is_nan_bits(Z[n * D + d]);
  if (bad) {
    atomicAdd(count, 1ull);
    if (write) for (int d = 0; d < D; ++d) Z[n * D + d] = 0.0;
  }
  if (write) mask[n] = bad ? 0.0 : 1.0;
}

// Is there a NaN anywhere in the (N, D) block?  The common answer is no, and then this flat scan — 16-byte loads where the
// buffer is aligned, eight per thread in flight, grid-stride — is all an upload pays (bound by the one read of Z; the row-wise
// kernel above reads with a stride of D doubles per lane: 0.64 TB/s at D = 16).  Only data that does hold a NaN goes on to it.
// With `sums` the kernel also accumulates the content checksum mimo_host_checksum defines (include/mimo_hip.h): m = w ^ (w >> 32),
// sums[0] += m, sums[1] += (count - index) m, 64-bit integer atomics (order-free) — the same read of Z serves both.
template <bool SUM>
__global__ __launch_bounds__(kWG) void nan_any_kernel(const unsigned long long* __restrict__ Z, int64_t count, unsigned int* __restrict__ flag,
                                                      unsigned long long* __restrict__ sums) {
  typedef unsigned long long u2 __attribute__((ext_vector_type(2)));
  const int64_t tid = (int64_t)blockIdx.x * kWG + threadIdx.x, nthr = (int64_t)gridDim.x * kWG;
  unsigned long long worst = 0ull;        // max of the exponent + mantissa bits seen: > 0x7ff0.. <=> some element is a NaN
  unsigned long long sa = 0ull, sb = 0ull;
  const bool aligned = (reinterpret_cast<uintptr_t>(Z) & 15u) == 0;
  const int64_t pairs = aligned ? count / 2 : 0;
  const u2* Z2 = reinterpret_cast<const u2*>(Z);
  auto take = [&](unsigned long long w, int64_t e) {
    const unsigned long long a0 = w & 0x7fffffffffffffffull;
    worst = worst > a0 ? worst : a0;
    if constexpr (SUM) {
      const unsigned long long m = w ^ (w >> 32);
      sa += m;
      sb += m * (unsigned long long)(count - e);
    }
  };
  int64_t i = tid;
  for (; i + 7 * nthr < pairs; i += 8 * nthr) {
    u2 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = Z2[i + j * nthr];
#pragma unroll
    for (int j = 0; j < 8; ++j) { take(v[j].x, 2 * (i + j * nthr)); take(v[j].y, 2 * (i + j * nthr) + 1); }
  }
  for (; i < pairs; i += nthr) {
    const u2 v = Z2[i];
    take(v.x, 2 * i); take(v.y, 2 * i + 1);
  }
  for (int64_t e = 2 * pairs + tid; e < count; e += nthr) take(Z[e], e);     // the odd element, or everything of an unaligned buffer
  if (worst > 0x7ff0000000000000ull) atomicOr(flag, 1u);
  if constexpr (SUM) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { sa += __shfl_xor(sa, s); sb += __shfl_xor(sb, s); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&sums[0], sa); atomicAdd(&sums[1], sb); }
  }
}

hipError_t launch_nan_any(const double* Z, int64_t count, unsigned int* flag, int num_cu, hipStream_t stream, unsigned long long* sums) {
  if (count <= 0) return hipSuccess;
  int64_t g = (int64_t)num_cu * 8, need = (count / 2 + kWG - 1) / kWG;
  if (g > need) g = need;
  if (g < 1) g = 1;
  const unsigned long long* Zw = reinterpret_cast<const unsigned long long*>(Z);
  if (sums) hipLaunchKernelGGL(nan_any_kernel<true>, dim3((unsigned)g), dim3(kWG), 0, stream, Zw, count, flag, sums);
  else hipLaunchKernelGGL(nan_any_kernel<false>, dim3((unsigned)g), dim3(kWG), 0, stream, Zw, count, flag, sums);
  return hipGetLastError();
}

__global__ __launch_bounds__(kWG) void mask_labels_kernel(const int32_t* __restrict__ labels, const double* __restrict__ mask,
                                                          int32_t* __restrict__ out, int64_t N, int K,
                                                          unsigned long long* __restrict__ bad_counts) {
  const int64_t n = (int64_t)blockIdx.x * kWG + threadIdx.x;
  if (n >= N) return;
  const int32_t l = labels[n];
  const bool keep = mask[n] != 0.0;
  out[n] = keep ? l : -1;
  if (!keep && bad_counts && l >= 0 && l < K) atomicAdd(&bad_counts[l], 1ull);     // integer: order-free
}

__global__ __launch_bounds__(kWG) void mask_table_kernel(const double* __restrict__ table, const double* __restrict__ mask,
                                                         double* __restrict__ out, int K, int64_t N) {
  const int64_t n = (int64_t)blockIdx.x * kWG + threadIdx.x;
  if (n >= N) return;
  const double m = mask[n];
  for (int k = 0; k < K; ++k) out[(int64_t)k * N + n] = table[(int64_t)k * N + n] * m;
}

// out[k][j] = table[k][cols[j]]: the few columns of a (K, N) table a host-side correction needs (rows with NaN)
__global__ __launch_bounds__(kWG) void gather_columns_kernel(const double* __restrict__ table, int K, int64_t N, const int64_t* __restrict__ cols,
                                                             int64_t ncols, double* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * kWG + threadIdx.x;
  if (e >= (int64_t)K * ncols) return;
  const int64_t k = e / ncols, j = e - k * ncols, n = cols[j];
  out[e] = (n >= 0 && n < N) ? table[k * N + n] : 0.0;
}
hipError_t launch_gather_columns(const double* table, int K, int64_t N, const int64_t* cols, int64_t ncols, double* out, hipStream_t stream) {
  if (ncols <= 0 || K <= 0) return hipSuccess;
  hipLaunchKernelGGL(gather_columns_kernel, dim3((unsigned)(((int64_t)K * ncols + kWG - 1) / kWG)), dim3(kWG), 0, stream, table, K, N, cols, ncols, out);
  return hipGetLastError();
}

hipError_t launch_nan_scan(double* Z, int64_t N, int D, double* mask, unsigned long long* count, bool write, hipStream_t stream) {
  if (N <= 0) return hipSuccess;
  hipLaunchKernelGGL(nan_scan_kernel, dim3((unsigned)((N + kWG - 1) / kWG)), dim3(kWG), 0, stream, Z, N, D, mask, count, write ? 1 : 0);
  return hipGetLastError();
}
hipError_t launch_mask_labels(const int32_t* labels, const double* mask, int32_t* out, int64_t N, int K,
                              unsigned long long* bad_counts, hipStream_t stream) {
  if (N <= 0) return hipSuccess;
  hipLaunchKernelGGL(mask_labels_kernel, dim3((unsigned)((N + kWG - 1) / kWG)), dim3(kWG), 0, stream, labels, mask, out, N, K, bad_counts);
  return hipGetLastError();
}
hipError_t launch_mask_table(const double* table, const double* mask, double* out, int K, int64_t N, hipStream_t stream) {
  if (N <= 0) return hipSuccess;
  hipLaunchKernelGGL(mask_table_kernel, dim3((unsigned)((N + kWG - 1) / kWG)), dim3(kWG), 0, stream, table, mask, out, K, N);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// reduce_partials + unpack_stats in ONE launch (they were 17.7 + 5.3 us behind a 57 us kernel at the C1 shape, with a launch
// gap between them): 64 consecutive elements of the partial blocks per workgroup, the G blocks split into 16 contiguous slices
// (one per wave of a 1024-thread workgroup — the two-launch version gave 4 waves 192 dependent steps each at G = 768), every
// slice summed in 4 interleaved chains, the slices combined as a fixed tree: the association depends on G only, never on
// timing.  The thread that holds the total of element (k, f) writes it where unpack_stats put it (packed S[K][1 + D + D^2],
// symmetric copies included); the workgroup that holds the four scalar slots writes scalars[3].
// ------------------------------------------------------------------------------------------
constexpr int kRuWG = 1024, kRuSlices = kRuWG / 64;
__global__ __launch_bounds__(kRuWG) void reduce_unpack_kernel(const double* __restrict__ partials, int G, int64_t stride,
                                                             const uint8_t* __restrict__ feat, int K, int D, int F, int F16,
                                                             double* __restrict__ S, double* __restrict__ scalars, int mask_structure) {
  __shared__ double part[kRuSlices][64];
  __shared__ double fin[64];
  const int ex = threadIdx.x & 63, gs = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + ex;
  const int g0 = (int)((int64_t)G * gs / kRuSlices), g1 = (int)((int64_t)G * (gs + 1) / kRuSlices);
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
  if (gs != 0) return;
  double v = 0.0;
  {
    double t[kRuSlices];
#pragma unroll
    for (int i = 0; i < kRuSlices; ++i) t[i] = part[i][ex];
#pragma unroll
    for (int w = kRuSlices / 2; w >= 1; w >>= 1)
#pragma unroll
      for (int i = 0; i < w; ++i) t[i] = t[2 * i] + t[2 * i + 1];
    v = t[0];
  }
  fin[ex] = v;
  const int Kpad = (K + 15) / 16 * 16;
  const int64_t nfeat = (int64_t)Kpad * F16;
  if (S && e < nfeat) {
    const int k = (int)(e / F16), f = (int)(e - (int64_t)k * F16);
    if (k < K && f < F) {
      const int aa = feat[2 * f], bb = feat[2 * f + 1];
      if (mask_structure == 1 && aa != bb && bb != D) v = 0.0;     // (small-shape kernel under a structure hint: see unpack_stats)
      if (mask_structure == 2 && bb != D) v = 0.0;
      double* Sk = S + (int64_t)k * (1 + D + D * D);
      if (aa == D) Sk[0] = v;
      else if (bb == D) Sk[1 + aa] = v;
      else { Sk[1 + D + aa * D + bb] = v; Sk[1 + D + bb * D + aa] = v; }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (one wave from here on: its own LDS stores are visible to it in order)
  if (scalars && e == nfeat) {
    const double slse = fin[ex], srl = fin[ex + 1];
    const bool split = fin[ex + 2] > 0.0;
    const long long nan_bits = 0x7ff8000000000000LL;
    scalars[0] = slse;
    scalars[1] = __longlong_as_double(split ? __double_as_longlong(srl) : nan_bits);
    scalars[2] = __longlong_as_double(split ? __double_as_longlong(slse - srl) : nan_bits);
  }
}

hipError_t launch_reduce_unpack(const double* partials, int G, int64_t stride, const uint8_t* feat, int K, int D, int F, int F16,
                                double* S_packed, double* scalars3, hipStream_t stream, int mask_structure) {
  if (S_packed && F < feat_count(D)) {   // reduced feature map: the entries outside it are zeros
    hipError_t e = hipMemsetAsync(S_packed, 0, sizeof(double) * (size_t)K * (1 + D + (size_t)D * D), stream);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(reduce_unpack_kernel, dim3((unsigned)((stride + 63) / 64)), dim3(kRuWG), 0, stream, partials, G, stride, feat,
                     K, D, F, F16, S_packed, scalars3, mask_structure);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Shader clock under float64 load: every workgroup runs a short v_fma_f64 loop bracketed by s_memtime (shader clock
// counter) and s_memrealtime (constant 100 MHz); clock = d memtime / d memrealtime x 100 MHz (as tools/f64_rates.hip).
// bench.py calls it right after its sustained leg, while the device is warm.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWG) void clock_probe_kernel(unsigned long long* __restrict__ out, int iters, double a, double b) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  double x0 = a + threadIdx.x * 1e-9, x1 = a * 2, x2 = a * 3, x3 = a * 4;
  for (int it = 0; it < iters; ++it) {
    asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b), "v"(a));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = t1 - t0;
    out[2 * blockIdx.x + 1] = r1 - r0;
  }
  if (x0 + x1 + x2 + x3 == 12345.678) out[0] = 0;      // keeps the loop
}

hipError_t launch_clock_probe(unsigned long long* out, int grid, int iters, hipStream_t stream) {
  hipLaunchKernelGGL(clock_probe_kernel, dim3(grid), dim3(kWG), 0, stream, out, iters, 0.999, 1e-3);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------
// components per lane / lanes per row for (Dz, K): KL = 4 up to Dz = 3, 2 at Dz = 4 (Theta rows + accumulators of a
// lane: 2 KL F doubles in registers); G = smallest power of two with G KL >= K
int small_kl(int D, int K) { return (D <= 2 || (D == 3 && K <= 4)) ? 4 : 2; }
int small_g(int D, int K) {
  const int kl = small_kl(D, K);
  int g = 1;
  while (g * kl < K) g <<= 1;
  return g;
}
// measured crossover against the MFMA tile kernels (tools/small_sweep.py, N = 1e7): Dz <= 2 up to K = 32; Dz = 3, 4 up
// to K = 16 (at K = 32 the tile kernels win: 1.04 / 1.13 ms against 1.01 / 1.20 ms, Gibbs 1.30 / 1.37 against 1.52 / 1.73)
bool small_covers(int D, int K) {
  return D >= 1 && D <= kSmallMaxD && K >= 1 && K <= (D <= 2 ? kSmallMaxK : 16) && small_g(D, K) <= 16;
}

typedef void (*small_fn)(const KernelArgs);

template <int DZ, int KL, int G>
static small_fn pick_small_mode(int mode) {
  switch (mode) {
    case kFastVI: return small_kernel<DZ, KL, G, kFastVI>;
    case kFastGibbs: return small_kernel<DZ, KL, G, kFastGibbs>;
    case kGeneric: return small_kernel<DZ, KL, G, kGeneric>;
    case kModeWeights: return small_kernel<DZ, KL, G, kModeWeights>;
    case kModeLabels: return small_kernel<DZ, KL, G, kModeLabels>;
  }
  return nullptr;
}
template <int DZ, int KL>
static small_fn pick_small_g(int G, int mode) {
  switch (G) {
    case 1: return pick_small_mode<DZ, KL, 1>(mode);
    case 2: return pick_small_mode<DZ, KL, 2>(mode);
    case 4: return pick_small_mode<DZ, KL, 4>(mode);
    case 8: return pick_small_mode<DZ, KL, 8>(mode);
    case 16: if constexpr (KL == 2) return pick_small_mode<DZ, KL, 16>(mode); else return nullptr;
    default: break;
  }
  return nullptr;
}
static small_fn resolve_small(const KernelArgs& a, int src) {
  int mode = src == kSrcWeights ? kModeWeights : src == kSrcLabels ? kModeLabels : kGeneric;
  if (src == kSrcEstep && a.do_stats && !a.split && !a.logp && !a.resp && !a.lse && (a.gibbs || !a.u))
    mode = a.gibbs ? kFastGibbs : kFastVI;
  const int G = small_g(a.D, a.K);
  switch (a.D) {
    case 1: return pick_small_g<1, 4>(G, mode);
    case 2: return pick_small_g<2, 4>(G, mode);
    case 3: return G == 1 ? pick_small_mode<3, 4, 1>(mode) : pick_small_g<3, 2>(G, mode);
    case 4: return pick_small_g<4, 2>(G, mode);
  }
  return nullptr;
}

int small_grid(const KernelArgs& a, int num_cu, int src) {
  int per_cu = 2;
  if (small_fn fn = resolve_small(a, src)) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(fn), kWG, 0) == hipSuccess && nb > 0)
      per_cu = nb;
    (void)hipGetLastError();
  }
  int cap = 4;
  if (const char* e = getenv("MIMO_SMALL_WG_PER_CU")) cap = atoi(e) > 0 ? atoi(e) : cap;   // tuning knob
  if (per_cu > cap) per_cu = cap;
  const int G = small_g(a.D, a.K), U = (a.D <= 2 && G == 1) ? MIMO_SMALL_U1 : 2;
  const int64_t rows_per_wg = (int64_t)4 * U * (64 / G);
  int64_t g = (int64_t)num_cu * per_cu, need = (a.N + rows_per_wg - 1) / rows_per_wg;
  if (g > need) g = need;
  if (g < 1) g = 1;
  return (int)g;
}

hipError_t launch_small(const KernelArgs& a, int src, int grid, hipStream_t stream, bool* unsupported) {
  *unsupported = false;
  small_fn fn = resolve_small(a, src);
  if (!fn) { *unsupported = true; return hipSuccess; }
  hipLaunchKernelGGL(fn, dim3(grid), dim3(kWG), 0, stream, a);
  return hipGetLastError();
}

}  // namespace mimo
