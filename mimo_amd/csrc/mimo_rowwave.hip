// gfx950 kernels of the Gibbs sweep at Dz <= 9, K <= 256 (BASELINE config C3: DP-GMM with Kmax = 256, D = 8):
//
//   gibbs_rowwave_kernel<KB>   label pass:  l = Theta . Phi'  ->  inverse-CDF draw, nothing but the labels leaves
//   label_stats_kernel<DZ>     statistics of the labels just drawn, bound by HBM (the data once + 4 bytes per row)
//
// Why not the fused tile kernel (mimo_kernels.hip, RBW = 4) for this shape: there a workgroup shares one 32-row tile,
// the l tile goes through LDS twice (66 KB per workgroup), four barriers per tile, and the 96 accumulator registers
// of the statistics push the kernel into scratch (C3: 8.1 ms, 42 % of the float64 matrix peak).  Here the roles are
// swapped — "row-owner" waves:
//
//   * Theta (K x F16, 98 KB at K = 256, Dz = 8) is loaded into LDS ONCE per workgroup and only read afterwards;
//   * a wave owns 16 data rows per step: B operand Phi[row j][4s + q] is built on the fly from the wave's own z rows
//     (2 LDS reads + 1 product per contraction step, shared by the K/16 MFMAs of the step), A operand = Theta slices
//     straight from LDS, accumulators = the l values of ALL components of the wave's 16 rows (4 K/16 doubles per lane);
//   * the components are permuted in the operand image so that lane (q, j) ends up with a CONTIGUOUS quarter of the
//     components of row j: max, exp, cumulative sums and the inverse-CDF count run in registers, three cross-lane
//     steps per row, no LDS round trip and no workgroup barrier anywhere in the loop.
//
// The statistics of hard labels are a scatter (S[label_n] += phi(z_n)); done deterministically without float atomics:
// per 512-row tile a bitmap per component (integer atomic OR: order-free), stable ranks by popcount, then thread k
// walks ITS rows in ascending order and accumulates the F features in registers across all tiles of the workgroup.
//
// Reference behaviour reproduced: mimo/mixtures/gmm.py:227-237 (resample_labels + resample_components' statistics),
// mimo/utils/stats.py:8-21 (label = #{k : u cum_K > cum_k}), gaussian.py:491-502, data.py:160-169.
#include "mimo_device.h"

#include <cstdlib>
#include <type_traits>

namespace mimo {

// diagnostic builds (-DMIMO_STAMPS, make stamps): cycles per phase of the label-statistics kernels, summed per wave
#ifdef MIMO_STAMPS
#define LS_STAMP_INIT unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t0 = __builtin_amdgcn_s_memtime();
#define LS_STAMP(i) { const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); st_[i] += t1_ - st_t0; st_t0 = t1_; }
#define LS_STAMP_STORE if (a.stamps && (threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 8; ++i_) a.stamps[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + i_] = st_[i_]; }
#else
#define LS_STAMP_INIT
#define LS_STAMP(i)
#define LS_STAMP_STORE
#endif

// ------------------------------------------------------------------------------------------
// Label pass.  KB = row blocks (16 components each) the accumulators cover; K <= 16 KB.
// Operand image (host, upload_theta_rowwave): slice e = s KB + rb, lane (i = lane & 15, kk = lane >> 4) holds
// Theta[comp(i, rb)][4 s + kk] with comp(i, rb) = (i & 3) V + 4 rb + (i >> 2), V = 4 KB: output lane (q, j)
// register r of row block rb (= A-row q + 4 r) is component q V + 4 rb + r of data row j.
// ------------------------------------------------------------------------------------------
constexpr int kRowWaveWG = 512;     // 8 wavefronts: two per SIMD (the accumulators need 8 KB VGPRs per lane)

template <int KB, int NS4>
__global__ __launch_bounds__(kRowWaveWG, 1) void gibbs_rowwave_kernel(const KernelArgs a) {
  constexpr int V = 4 * KB;          // components per lane
  constexpr int NCH = KB / 2;        // chunks of 8 components (two row blocks) for the cumulative sums
  static_assert(KB % 2 == 0 && KB >= 2 && KB <= 16, "row blocks per wave");
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int NS = 4 * NS4;                        // contraction steps (F16 / 4): compile-time, the step loop is straight-line
  const int ZS = a.ZS;
  double* Th = reinterpret_cast<double*>(smem);      // [(NS KB + 4)][64]   (+4: the operand prefetch runs past the end)
  double* etab = Th + (size_t)(NS * KB + 4) * 64;    // [kExpTab]
  double* Zall = etab + kExpTab;                     // [8 waves][16][ZS]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, j = lane & 15;
  const int D = a.D, K = a.K;
  const int64_t N = a.N;
  double* Zw = Zall + (size_t)wave * 16 * ZS;

  for (int e = tid; e < NS * KB * 64; e += kRowWaveWG) Th[e] = a.theta[e];
  for (int e = tid; e < 4 * 64; e += kRowWaveWG) Th[NS * KB * 64 + e] = 0.0;
  for (int e = tid; e < kExpTab; e += kRowWaveWG) etab[e] = exp_tab_entry_c(e);
  // a.fuse_hist: the histogram of the labels this launch draws, for the slot table of label_stats_slots_kernel — 16 LDS atomics per
  // wave step here instead of a pass of its own over the labels (label_hist_kernel: 36 us at N = 1e7, same-address atomics of skewed labels)
  __shared__ uint32_t hloc[256];
  if (tid < 256) hloc[tid] = 0u;
  wg_sync();

  // z rows of a 16-row step: element e = lane + 64 i of the (16, D) block, i < ZI (16 D <= 144 up to Dz = 9, <= 256 up to 16)
  constexpr int ZI = 4;     // (16 Dz <= 256: Dz <= 16 — full maps beyond Dz = 9 and the reduced maps of structured blocks)
  const int64_t nsteps = (N + 15) / 16;
  const int64_t nwaves = (int64_t)gridDim.x * (kRowWaveWG / 64), wv = (int64_t)blockIdx.x * (kRowWaveWG / 64) + wave;
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

  const double* zrow = Zw + j * ZS;
  const double* thl = Th + lane;
  // B operand of step s, lane (q, j): feature 4 s + q of row j = z~[a] z~[b]; the two LDS addresses are fixed for the
  // whole kernel (2 NS registers), so a step costs two LDS reads and one product; beyond 16 steps (Dz > 9) the two
  // byte offsets share one register per step (two more integer instructions per step)
  constexpr bool kPacked = NS > 16;
  const double* fpa[kPacked ? 1 : NS];
  const double* fpb[kPacked ? 1 : NS];
  uint32_t fo[kPacked ? NS : 1];
#pragma unroll
  for (int s2 = 0; s2 < NS; ++s2) {
    if constexpr (kPacked) {
      fo[s2] = 8u * a.feat[2 * (4 * s2 + q)] | (8u * a.feat[2 * (4 * s2 + q) + 1]) << 16;
    } else {
      fpa[s2] = zrow + a.feat[2 * (4 * s2 + q)];
      fpb[s2] = zrow + a.feat[2 * (4 * s2 + q) + 1];
    }
  }
  auto feature = [&](int s2) -> double {
    if constexpr (kPacked) {
      const char* zb = reinterpret_cast<const char*>(zrow);
      return *reinterpret_cast<const double*>(zb + (fo[s2] & 0xffffu)) * *reinterpret_cast<const double*>(zb + (fo[s2] >> 16));
    } else {
      return *fpa[s2] * *fpb[s2];
    }
  };
  // Philox uniforms four steps at a time: lane (q, j) draws the uniform of row j of this wave's step t + q nwaves
  double ubatch = 0.0;
  int uphase = 0;

  for (int64_t t = wv; t < nsteps; t += nwaves) {
    const int64_t n = t * 16 + j;
    const bool valid = n < N;
    // ---- stage this step's z~ rows in the wave's own LDS block (LDS operations of one wave execute in order)
#pragma unroll
    for (int i = 0; i < ZI; ++i)
      if (zoff[i] >= 0) Zw[zoff[i]] = zr[i];
    if (q == 0) {
      Zw[j * ZS + D] = valid ? 1.0 : 0.0;     // rows past N: every feature 0, l = 0, never written
      Zw[j * ZS + D + 1] = 0.0;               // padded features read this slot
    }
    if (t + nwaves < nsteps) load_z(t + nwaves);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---- L (16 KB x 16) = Theta . Phi' ---------------------------------------------------------------
    d4 acc[KB];
#pragma unroll
    for (int rb = 0; rb < KB; ++rb) acc[rb] = d4{0.0, 0.0, 0.0, 0.0};
    constexpr int PF = 4;              // Theta slices in flight; slice e = s KB + rb sits in slot e % PF
    double ring[PF];
#pragma unroll
    for (int e = 0; e < PF; ++e) ring[e] = thl[e * 64];
    double bq = feature(0);
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {            // NS is a multiple of 4: slot (s2 KB + rb) % PF is static
      const double bcur = bq;
      if (s2 + 1 < NS) bq = feature(s2 + 1);
#pragma unroll
      for (int rb = 0; rb < KB; ++rb) {
        const int e = s2 * KB + rb;
        const double av = ring[e % PF];
        ring[e % PF] = thl[(e + PF) * 64];       // (the last step reads the 4 zero slices behind the image)
        acc[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bcur, acc[rb], 0, 0, 0);
      }
    }

    // ---- draw: lane (q, j) holds components q V .. q V + V - 1 of row j, x[4 rb + r] = acc[rb][r] ----------
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
    // e = exp(l - max), then inclusive cumulative sums inside chunks of 8 (independent chains across the chunks)
    double base[NCH + 1];
    base[0] = 0.0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      double x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = exp_nonpos_t2048c(acc[2 * c + (i >> 2)][i & 3] - m, etab);
#pragma unroll
      for (int i = 1; i < 8; ++i) x[i] += x[i - 1];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[2 * c + (i >> 2)][i & 3] = x[i];
      base[c + 1] = x[7];
      __builtin_amdgcn_sched_barrier(0);       // one chunk of exp chains in flight at a time (register pressure)
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) base[c + 1] += base[c];      // base[c] = sum of the chunks before c
    const double cum = base[NCH];
    double incl = cum;                          // inclusive scan over the four quarters of the row (lanes j, j+16, ..)
    {
      double v = __shfl_up(incl, 16);  if (q >= 1) incl += v;
      v = __shfl_up(incl, 32);         if (q >= 2) incl += v;
    }
    double excl = __shfl_up(incl, 16);
    if (q == 0) excl = 0.0;
    const double ctot = __shfl(incl, 48 + j);
    double uu;
    if (a.u) {
      uu = valid ? a.u[n] : 0.0;
    } else {
      if (uphase == 0)        // (wave-uniform) the same counters as one draw per step: the labels are the same labels
        ubatch = philox_uniform(a.seed, (uint64_t)(a.row0 + (t + (int64_t)q * nwaves) * 16 + j), a.sweep);
      uu = __shfl(ubatch, uphase * 16 + j);
      uphase = (uphase + 1) & 3;
    }
    const double tl = uu * ctot - excl;
    int cnt = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const double tc = tl - base[c];
#pragma unroll
      for (int i = 0; i < 8; ++i) cnt += tc > acc[2 * c + (i >> 2)][i & 3] ? 1 : 0;
    }
    cnt += __shfl_xor(cnt, 16);
    cnt += __shfl_xor(cnt, 32);
    const int label = cnt < K ? cnt : K - 1;
    if (q == 0 && valid) {
      a.labels[n] = label;
      if (a.fuse_hist) atomicAdd(&hloc[label], 1u);
    }
    __builtin_amdgcn_s_setprio(0);
  }
  if (a.fuse_hist) {                       // (wave-uniform)
    wg_sync();
    if (tid < 256 && hloc[tid]) atomicAdd(&a.aux[tid], hloc[tid]);
  }
}

// ------------------------------------------------------------------------------------------
// The same label pass where Theta does not fit LDS (K F16 8 bytes: 327 KB at K = 256, Dz = 16; 590 KB at K = 128, Dz = 32):
// the operand image STREAMS through a double buffer in chunks of NSC contraction steps (48 KB), the workgroup's eight waves
// walk the chunks together — one barrier per chunk, i.e. per NSC KB = 96 matrix instructions of every wave — and the next
// chunk (cyclic: Theta does not change from step to step) is fetched from L2 into registers under the current chunk's
// products and stored behind them.  The accumulators carry the l values of all components across the chunks, the draw is the
// one of gibbs_rowwave_kernel.  The feature pair of a step comes from a small LDS table (NS x 4 packed byte offsets) instead of
// 2 NS address registers.  These shapes ran on the tile kernels' pipelined E-step (wide_estep_kernel: K = 128, Dz = 16 at 49 % of
// the FP64 rate in the label pass against 74 % for the row-owner kernel at K = 64).
// L2 -> LDS traffic: the whole image per 128 rows (Dz = 16, K = 256: 2.5 KB per row, 25 GB per 1e7 rows — a quarter of the
// pass's matrix time at the 10 TB/s the L2s deliver, and hidden under it).
// ------------------------------------------------------------------------------------------
#ifndef MIMO_STREAM_TAIL
// 0: all padded steps computed, 1: a (wave-uniform) guard per step, 2: the guarded loop for a partial last chunk only.  One box,
// N = 2e6, label pass + statistics in us (profiles/r03_stream_tail_variants.txt), variants 0 / 1 / 2: Dz=16, K=128 1924 / 1746 / 1807;
// Dz=20, K=64 1653 / 1521 / 1549; Dz=24, K=64 2143 / 2020 / 2071; Dz=16, K=256 3381 / 3278 / 6364 (the two loop bodies of variant 2
// spill there); shapes without padding Dz=32, K=128 6232 / 6180 / 6476, Dz=12, K=192 1992 / 2050 / 2184
#define MIMO_STREAM_TAIL 1
#endif
constexpr int stream_nsc(int KB) { return KB <= 2 ? 48 : KB <= 4 ? 24 : KB <= 6 ? 16 : KB <= 8 ? 12 : KB <= 14 ? 8 : 6; }   // (NSC KB: a multiple of 16)

template <int KB, int ZI>
__global__ __launch_bounds__(kRowWaveWG, 1) void gibbs_stream_kernel(const KernelArgs a, int NSP, int NS) {
  constexpr int V = 4 * KB, NCH = KB / 2, NSC = stream_nsc(KB);
  constexpr int CH = NSC * KB * 64;                   // doubles per chunk
  constexpr int NLD = CH / (2 * kRowWaveWG);          // 16-byte loads per thread and chunk
  static_assert(KB % 2 == 0 && KB >= 2 && KB <= 16 && CH % (2 * kRowWaveWG) == 0, "row blocks per wave / chunk geometry");
  typedef double d2 __attribute__((ext_vector_type(2)));
  extern __shared__ __align__(16) unsigned char smem[];
  const int ZS = a.ZS;
  double* buf = reinterpret_cast<double*>(smem);      // [2][CH]
  double* etab = buf + 2 * CH;                        // [kExpTab]
  double* Zall = etab + kExpTab;                      // [8 waves][16][ZS]
  uint32_t* ftab = reinterpret_cast<uint32_t*>(Zall + (size_t)(kRowWaveWG / 64) * 16 * ZS);   // [NSP][4] packed byte offsets of a step's feature pair

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, j = lane & 15;
  const int D = a.D, K = a.K;
  const int64_t N = a.N;
  const int nch = NSP / NSC;                          // chunks per pass over the image (host: NSP is a multiple of NSC)
  double* Zw = Zall + (size_t)wave * 16 * ZS;

  for (int e = tid; e < kExpTab; e += kRowWaveWG) etab[e] = exp_tab_entry_c(e);
  for (int e = tid; e < NSP * 4; e += kRowWaveWG) {
    const int f = e;                                  // feature 4 s + q; beyond the table: the zero slot of z~
    const uint32_t fa = f < a.F16 ? a.feat[2 * f] : (uint32_t)(D + 1), fb = f < a.F16 ? a.feat[2 * f + 1] : (uint32_t)(D + 1);
    ftab[e] = 8u * fa | (8u * fb) << 16;
  }
  {                                                   // chunk 0 -> buffer 0 (the loop's first barrier publishes it)
    const d2* src = reinterpret_cast<const d2*>(a.theta) + tid;
    d2* dst = reinterpret_cast<d2*>(buf) + tid;
#pragma unroll
    for (int i = 0; i < NLD; ++i) dst[i * kRowWaveWG] = src[i * kRowWaveWG];
  }

  const int64_t nsteps = (N + 127) / 128;             // workgroup steps: 8 waves x 16 rows
  int zoff[ZI];
#pragma unroll
  for (int i = 0; i < ZI; ++i) {
    const int e = lane + 64 * i, r = e / D;
    zoff[i] = e < 16 * D ? r * ZS + (e - r * D) : -1;
  }
  double zr[ZI];
  auto load_z = [&](int64_t t) {
    const int64_t base = (t * 8 + wave) * 16 * D, total = N * D;
#pragma unroll
    for (int i = 0; i < ZI; ++i) {
      const int64_t gidx = base + lane + 64 * i;
      zr[i] = (zoff[i] >= 0 && gidx < total) ? a.Z[gidx] : 0.0;
    }
  };
  if ((int64_t)blockIdx.x < nsteps) load_z(blockIdx.x);

  const char* zb = reinterpret_cast<const char*>(Zw + j * ZS);
  auto feature = [&](int s) -> double {
    const uint32_t u = ftab[4 * s + q];
    return *reinterpret_cast<const double*>(zb + (u & 0xffffu)) * *reinterpret_cast<const double*>(zb + (u >> 16));
  };
  double ubatch = 0.0;
  int uphase = 0;
  int gc = 0;                                          // chunks done: chunk gc lives in buffer gc & 1

  for (int64_t t = blockIdx.x; t < nsteps; t += gridDim.x) {
    const int64_t n = (t * 8 + wave) * 16 + j;
    const bool valid = n < N;
#pragma unroll
    for (int i = 0; i < ZI; ++i)
      if (zoff[i] >= 0) Zw[zoff[i]] = zr[i];
    if (q == 0) {
      Zw[j * ZS + D] = valid ? 1.0 : 0.0;     // rows past N: every feature 0, l = 0, never written
      Zw[j * ZS + D + 1] = 0.0;
    }
    if (t + gridDim.x < nsteps) load_z(t + gridDim.x);

    d4 acc[KB];
#pragma unroll
    for (int rb = 0; rb < KB; ++rb) acc[rb] = d4{0.0, 0.0, 0.0, 0.0};
    for (int c = 0; c < nch; ++c, ++gc) {
      wg_sync();                               // chunk gc stands in its buffer; everybody is done with the other one
      const double* bc = buf + (gc & 1) * CH + lane;
      // the next chunk of the cyclic walk: global -> registers now, registers -> the other buffer behind the products
      d2 stg[NLD];
      {
        const int cn = c + 1 < nch ? c + 1 : 0;
        const d2* src = reinterpret_cast<const d2*>(a.theta + (size_t)cn * CH) + tid;
#pragma unroll
        for (int i = 0; i < NLD; ++i) stg[i] = src[i * kRowWaveWG];
      }
      constexpr int PF = 4;
      double ring[PF];
#pragma unroll
      for (int e = 0; e < PF; ++e) ring[e] = bc[e * 64];
      const int s0 = c * NSC;
      // steps of this chunk that carry features (wave-uniform): the image is padded to whole chunks with zero rows, and the
      // products of the padding — up to a fifth of the pass at Dz = 16, K = 96 / 128 or Dz = 20, K = 64 — are skipped
      const int lim = NS - s0;
      double bq = feature(s0);
      auto products = [&](auto guarded) {
#pragma unroll
        for (int s2 = 0; s2 < NSC; ++s2) {
          if (!decltype(guarded)::value || s2 < lim) {
            const double bcur = bq;
            if (s2 + 1 < NSC) bq = feature(s0 + s2 + 1);
#pragma unroll
            for (int rb = 0; rb < KB; ++rb) {
              const int e = s2 * KB + rb;
              const double av = ring[e % PF];
              if (e + PF < NSC * KB) ring[e % PF] = bc[(e + PF) * 64];
              acc[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bcur, acc[rb], 0, 0, 0);
            }
          }
        }
      };
#if MIMO_STREAM_TAIL == 0
      products(std::false_type{});
#elif MIMO_STREAM_TAIL == 1
      products(std::true_type{});
#else
      if (lim >= NSC) products(std::false_type{}); else products(std::true_type{});
#endif
      {
        d2* dst = reinterpret_cast<d2*>(buf + ((gc + 1) & 1) * CH) + tid;
#pragma unroll
        for (int i = 0; i < NLD; ++i) dst[i * kRowWaveWG] = stg[i];
      }
    }

    // ---- draw: lane (q, j) holds components q V .. q V + V - 1 of row j, x[4 rb + r] = acc[rb][r] (as gibbs_rowwave_kernel)
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
    double base[NCH + 1];
    base[0] = 0.0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      double x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = exp_nonpos_t2048c(acc[2 * c + (i >> 2)][i & 3] - m, etab);
#pragma unroll
      for (int i = 1; i < 8; ++i) x[i] += x[i - 1];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[2 * c + (i >> 2)][i & 3] = x[i];
      base[c + 1] = x[7];
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) base[c + 1] += base[c];
    const double cum = base[NCH];
    double incl = cum;
    {
      double v = __shfl_up(incl, 16);  if (q >= 1) incl += v;
      v = __shfl_up(incl, 32);         if (q >= 2) incl += v;
    }
    double excl = __shfl_up(incl, 16);
    if (q == 0) excl = 0.0;
    const double ctot = __shfl(incl, 48 + j);
    double uu;
    if (a.u) {
      uu = valid ? a.u[n] : 0.0;
    } else {
      if (uphase == 0)        // four workgroup steps at a time: lane (q, j) draws row j of this wave's step t + q gridDim.x
        ubatch = philox_uniform(a.seed, (uint64_t)(a.row0 + ((t + (int64_t)q * gridDim.x) * 8 + wave) * 16 + j), a.sweep);
      uu = __shfl(ubatch, uphase * 16 + j);
      uphase = (uphase + 1) & 3;
    }
    const double tl = uu * ctot - excl;
    int cnt = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const double tc = tl - base[c];
#pragma unroll
      for (int i = 0; i < 8; ++i) cnt += tc > acc[2 * c + (i >> 2)][i & 3] ? 1 : 0;
    }
    cnt += __shfl_xor(cnt, 16);
    cnt += __shfl_xor(cnt, 32);
    const int label = cnt < K ? cnt : K - 1;
    if (q == 0 && valid) a.labels[n] = label;
    __builtin_amdgcn_s_setprio(0);
  }
}

// contraction steps of the streamed image: F16 / 4 padded to whole chunks
int stream_ns_pad(int KB, int F16) { const int nsc = stream_nsc(KB), ns = F16 / 4; return (ns + nsc - 1) / nsc * nsc; }
size_t stream_lds_bytes(int KB, int F16, int ZS) {
  return sizeof(double) * ((size_t)2 * stream_nsc(KB) * KB * 64 + kExpTab + (size_t)(kRowWaveWG / 64) * 16 * ZS)
         + sizeof(uint32_t) * (size_t)stream_ns_pad(KB, F16) * 4;
}

size_t rowwave_lds_bytes(int KB, int NS, int ZS) {
  return sizeof(double) * ((size_t)(NS * KB + 4) * 64 + kExpTab + (size_t)(kRowWaveWG / 64) * 16 * ZS);
}

int rowwave_kb(int K) {           // row blocks the kernel is instantiated for: 2, 4, .., 16
  int kb = (K + 15) / 16;
  kb += kb & 1;
  return kb < 2 ? 2 : kb;
}

// Smallest K that takes the row-owner route (tuning knob; default: every K).  Measured against the fused tile kernels,
// N = 1e7 (tools/midk_time.py, labels + statistics): K = 64, D = 8 sweep 2.50 -> 1.69 ms, K = 32: 1.99 -> 1.05, K = 17: 1.94 ->
// 1.07, K = 16: 1.54 -> 1.05, K = 8: 1.52 -> 1.00, D = 5, K = 16: 1.52 -> 0.82, D = 7, K = 4: 1.43 -> 0.98 (the last three
// only since the label-statistics pass spreads a component's rows over 256 / Kp threads: with one thread per
// component they lost).
static int rowwave_min_k() {
  static const int v = [] { const char* e = getenv("MIMO_ROWWAVE_MIN_K"); return e ? atoi(e) : 1; }();
  return v;
}

// K <= 256 (from rowwave_min_k) at Dz <= 9 (F16 <= 64); Dz 10 .. 16 (F16 <= 160) while the operand image fits LDS next to
// the exp table: K <= 64, and K <= 128 up to Dz = 12 (the instantiations below)
static bool rowwave_resident(int K, int F16, int ZS) {
  if (K < rowwave_min_k() || K > 256 || F16 > 160) return false;
  const int kb = rowwave_kb(K);
  if (F16 > 64 && !(kb <= 4 || (kb <= 8 && F16 <= 96))) return false;
  return rowwave_lds_bytes(kb, F16 / 4, ZS) + 1024 <= 160 * 1024;      // (+ the kernel's static 1 KB label histogram)
}
// Row blocks for the shape: rowwave_kb(K), except that the streamed walk of 193 <= K <= 224 (14 row blocks: chunks of 8 steps = 57 KB,
// two of them + the z rows of Dz >= 25 do not fit 160 KB) runs with 16 (chunks of 6 steps = 49 KB) — 14 % more matrix work, no table
int rowwave_kb_shape(int K, int F16, int ZS) {
  const int kb = rowwave_kb(K);
  if (kb == 14 && !rowwave_resident(K, F16, ZS) && stream_lds_bytes(14, F16, ZS) > 160 * 1024) return 16;
  return kb;
}
// ... and beyond that, up to Dz = 32 (F16 <= 576), with Theta streamed through LDS (MIMO_ROWWAVE_STREAM=0: off, tuning knob)
static bool rowwave_streams(int K, int F16, int ZS) {
  static const bool on = [] { const char* e = getenv("MIMO_ROWWAVE_STREAM"); return !e || atoi(e) != 0; }();
  if (!on || K < rowwave_min_k() || K > 256 || F16 > 576 || rowwave_resident(K, F16, ZS)) return false;
  return stream_lds_bytes(rowwave_kb_shape(K, F16, ZS), F16, ZS) <= 160 * 1024;
}
bool rowwave_covers(int K, int F16, int ZS) { return rowwave_resident(K, F16, ZS) || rowwave_streams(K, F16, ZS); }
// contraction steps the operand image must hold for (K, F16): F16 / 4, or whole chunks of the streamed walk
int rowwave_image_ns(int K, int F16, int ZS) {
  return rowwave_streams(K, F16, ZS) ? stream_ns_pad(rowwave_kb_shape(K, F16, ZS), F16) : F16 / 4;
}

typedef void (*rowwave_fn)(const KernelArgs);
template <int NS4>
static rowwave_fn pick_rowwave_kb(int kb) {
  switch (kb) {
    case 2: return gibbs_rowwave_kernel<2, NS4>;
    case 4: return gibbs_rowwave_kernel<4, NS4>;
    case 6: return gibbs_rowwave_kernel<6, NS4>;
    case 8: return gibbs_rowwave_kernel<8, NS4>;
    case 10: return gibbs_rowwave_kernel<10, NS4>;
    case 12: return gibbs_rowwave_kernel<12, NS4>;
    case 14: return gibbs_rowwave_kernel<14, NS4>;
    case 16: return gibbs_rowwave_kernel<16, NS4>;
  }
  return nullptr;
}
template <int NS4>
static rowwave_fn pick_rowwave_wide(int kb) {      // Dz 10 .. 16: K <= 64 (and K <= 128 while F16 <= 96)
  switch (kb) {
    case 2: return gibbs_rowwave_kernel<2, NS4>;
    case 4: return gibbs_rowwave_kernel<4, NS4>;
    case 6: if constexpr (NS4 <= 6) return gibbs_rowwave_kernel<6, NS4>; else return nullptr;
    case 8: if constexpr (NS4 <= 6) return gibbs_rowwave_kernel<8, NS4>; else return nullptr;
  }
  return nullptr;
}
static rowwave_fn pick_rowwave(int kb, int F16) {
  switch (F16 / 16) {
    case 1: return pick_rowwave_kb<1>(kb);
    case 2: return pick_rowwave_kb<2>(kb);
    case 3: return pick_rowwave_kb<3>(kb);
    case 4: return pick_rowwave_kb<4>(kb);
    case 5: return pick_rowwave_wide<5>(kb);
    case 6: return pick_rowwave_wide<6>(kb);
    case 7: return pick_rowwave_wide<7>(kb);
    case 8: return pick_rowwave_wide<8>(kb);
    case 9: return pick_rowwave_wide<9>(kb);
    case 10: return pick_rowwave_wide<10>(kb);
  }
  return nullptr;
}

int rowwave_grid(const KernelArgs& a, int num_cu) {
  const int64_t steps = (a.N + 15) / 16, need = (steps + 7) / 8;
  int64_t g = num_cu;
  if (g > need) g = need;
  return (int)(g < 1 ? 1 : g);
}

typedef void (*stream_fn)(const KernelArgs, int, int);
template <int ZI>
static stream_fn pick_stream(int kb) {
  switch (kb) {
    case 2: return gibbs_stream_kernel<2, ZI>;    case 4: return gibbs_stream_kernel<4, ZI>;
    case 6: return gibbs_stream_kernel<6, ZI>;    case 8: return gibbs_stream_kernel<8, ZI>;
    case 10: return gibbs_stream_kernel<10, ZI>;  case 12: return gibbs_stream_kernel<12, ZI>;
    case 14: return gibbs_stream_kernel<14, ZI>;  case 16: return gibbs_stream_kernel<16, ZI>;
  }
  return nullptr;
}

hipError_t launch_gibbs_rowwave(const KernelArgs& a, int grid, hipStream_t stream) {
  if (rowwave_streams(a.K, a.F16, a.ZS)) {
    const int kb = rowwave_kb_shape(a.K, a.F16, a.ZS);
    stream_fn fn = 16 * a.D <= 256 ? pick_stream<4>(kb) : pick_stream<8>(kb);
    if (!fn || a.D > 32) return hipErrorInvalidValue;
    const size_t lds = stream_lds_bytes(kb, a.F16, a.ZS);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int64_t need = (a.N + 127) / 128;
    int g = grid;
    if (g > need) g = (int)(need < 1 ? 1 : need);
    hipLaunchKernelGGL(fn, dim3(g), dim3(kRowWaveWG), lds, stream, a, stream_ns_pad(kb, a.F16), a.F16 / 4);
    return hipGetLastError();
  }
  const int kb = rowwave_kb(a.K);
  rowwave_fn fn = pick_rowwave(kb, a.F16);
  if (!fn) return hipErrorInvalidValue;
  const size_t lds = rowwave_lds_bytes(kb, a.F16 / 4, a.ZS);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(kRowWaveWG), lds, stream, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Mean-field / EM pass (softmax + weighted statistics) on row-owner waves: K <= 64, Dz <= 9.
// Through the tile kernels this regime runs at 44 - 60 % of its MFMA time (N = 1e7, D = 8: K = 64 2.58 ms against a
// floor of 1.56 ms, K = 32 1.90 ms): one 32-row tile per workgroup, four barriers per tile, the l / r tile through LDS
// for all four waves.  Here a wave owns 16 rows for BOTH products:
//   L = Theta . Phi'   exactly as in gibbs_rowwave_kernel (Theta in LDS, B operand built on the fly);
//   softmax over the lane's contiguous quarter of the components in registers;
//   the wave's r values go to a wave-private LDS block Rt[row][slot] (slot = 16 rb + 4 r + q: the A-operand order of
//   the second product), then  S += R . Phi :  A = Rt[row 4t + kk][16 rb + i], B = Phi[row 4t + kk][16 cb + j] built
//   on the fly from the same z rows; the wave keeps its OWN K x F16 statistic block in registers (KB NCB accumulator
//   quads: 96 VGPRs at K = 64, Dz = 8) across all its steps.  No workgroup barrier in the loop.
// At the end the 8 waves of the workgroup add their blocks through LDS in wave order, one accumulator quad at a time.
// S-row i of row block rb holds component (i & 3) V + 4 rb + (i >> 2) (the permutation of the operand image).
// ------------------------------------------------------------------------------------------
template <int KB, int NS4>
__global__ __launch_bounds__(kRowWaveWG, 1) void vi_rowwave_kernel(const KernelArgs a) {
  constexpr int V = 4 * KB, NS = 4 * NS4, NCB = NS4;
  constexpr int RS2 = 16 * KB + 8;                    // row stride of the wave's r block (doubles)
  static_assert(KB == 2 || KB == 4, "K <= 64");
  extern __shared__ __align__(16) unsigned char smem[];
  const int ZS = a.ZS;
  double* Th = reinterpret_cast<double*>(smem);       // [(NS KB + 4)][64]
  double* etab = Th + (size_t)(NS * KB + 4) * 64;     // [kExpTab]
  double* Zall = etab + kExpTab;                      // [8][16][ZS]
  double* Rall = Zall + (size_t)(kRowWaveWG / 64) * 16 * ZS;   // [8][16][RS2]
  double* sred = Rall + (size_t)(kRowWaveWG / 64) * 16 * RS2;  // [8]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, j = lane & 15;
  const int D = a.D, K = a.K;
  const int64_t N = a.N;
  double* Zw = Zall + (size_t)wave * 16 * ZS;
  double* Rt = Rall + (size_t)wave * 16 * RS2;

  for (int e = tid; e < NS * KB * 64; e += kRowWaveWG) Th[e] = a.theta[e];
  for (int e = tid; e < 4 * 64; e += kRowWaveWG) Th[NS * KB * 64 + e] = 0.0;
  for (int e = tid; e < kExpTab; e += kRowWaveWG) etab[e] = exp_tab_entry_c(e);
  wg_sync();

  const int64_t nsteps = (N + 15) / 16;
  const int64_t nwaves = (int64_t)gridDim.x * (kRowWaveWG / 64), wv = (int64_t)blockIdx.x * (kRowWaveWG / 64) + wave;
  constexpr int ZI = 4;      // 16 Dz <= 256 elements per step (Dz <= 16: the reduced feature maps of structured blocks)
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

  const double* zrow = Zw + j * ZS;
  const double* thl = Th + lane;
  constexpr bool kPacked = KB == 4 && NS > 12;        // (K > 32 at Dz = 9: the two offsets of a step share a register)
  const double* fpa[kPacked ? 1 : NS];
  const double* fpb[kPacked ? 1 : NS];
  uint32_t fo[kPacked ? NS : 1];
#pragma unroll
  for (int s2 = 0; s2 < NS; ++s2) {
    if constexpr (kPacked) {
      fo[s2] = 8u * a.feat[2 * (4 * s2 + q)] | (8u * a.feat[2 * (4 * s2 + q) + 1]) << 16;
    } else {
      fpa[s2] = zrow + a.feat[2 * (4 * s2 + q)];
      fpb[s2] = zrow + a.feat[2 * (4 * s2 + q) + 1];
    }
  }
  auto feature = [&](int s2) -> double {
    if constexpr (kPacked) {
      const char* zb = reinterpret_cast<const char*>(zrow);
      return *reinterpret_cast<const double*>(zb + (fo[s2] & 0xffffu)) * *reinterpret_cast<const double*>(zb + (fo[s2] >> 16));
    } else {
      return *fpa[s2] * *fpb[s2];
    }
  };
  // second product: lane (kk = q, col j) builds feature 16 cb + j of row 4 t + kk; A operand = Rt[4 t + kk][16 rb + i], i = j
  const double* zk = Zw + q * ZS;                     // row kk of the step's block; row 4 t + kk is 4 t ZS doubles further
  int spa[NCB], spb[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    spa[cb] = a.feat[2 * (16 * cb + j)];
    spb[cb] = a.feat[2 * (16 * cb + j) + 1];
  }
  const double* rk = Rt + q * RS2 + j;
  double* rw = Rt + j * RS2 + q;                      // this lane's r values: slot 16 rb + 4 r + q of row j

  d4 sacc[KB][NCB];
#pragma unroll
  for (int rb = 0; rb < KB; ++rb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) sacc[rb][cb] = d4{0.0, 0.0, 0.0, 0.0};
  double sc_lse = 0.0, sc_prod = 1.0;
  int since_flush = 0;

  for (int64_t t = wv; t < nsteps; t += nwaves) {
    const int64_t n = t * 16 + j;
    const bool valid = n < N;
#pragma unroll
    for (int i = 0; i < ZI; ++i)
      if (zoff[i] >= 0) Zw[zoff[i]] = zr[i];
    if (q == 0) {
      Zw[j * ZS + D] = valid ? 1.0 : 0.0;     // rows past N: every feature 0 — l = 0, and nothing reaches the statistics
      Zw[j * ZS + D + 1] = 0.0;
    }
    if (t + nwaves < nsteps) load_z(t + nwaves);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---- L = Theta . Phi' ------------------------------------------------------------------------
    d4 acc[KB];
#pragma unroll
    for (int rb = 0; rb < KB; ++rb) acc[rb] = d4{0.0, 0.0, 0.0, 0.0};
    {
      constexpr int PF = 4;
      double ring[PF];
#pragma unroll
      for (int e = 0; e < PF; ++e) ring[e] = thl[e * 64];
      double bq = feature(0);
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) {
        const double bcur = bq;
        if (s2 + 1 < NS) bq = feature(s2 + 1);
#pragma unroll
        for (int rb = 0; rb < KB; ++rb) {
          const int e = s2 * KB + rb;
          const double av = ring[e % PF];
          ring[e % PF] = thl[(e + PF) * 64];
          acc[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bcur, acc[rb], 0, 0, 0);
        }
      }
    }

    // ---- softmax over the row's components (this lane: components q V .. q V + V - 1) ------------------
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
    double sv[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int rb = 0; rb < KB; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[rb][r] = exp_nonpos_t2048c(acc[rb][r] - m, etab);
        sv[r] += acc[rb][r];
      }
    double ssum = (sv[0] + sv[1]) + (sv[2] + sv[3]);
    ssum += __shfl_xor(ssum, 16);
    ssum += __shfl_xor(ssum, 32);
    double inv = __builtin_amdgcn_rcp(ssum);
    inv = fma(fma(-ssum, inv, 1.0), inv, inv);
    inv = fma(fma(-ssum, inv, 1.0), inv, inv);
    if (q == 0 && valid) { sc_lse += m; sc_prod *= ssum; }
    // (rows past N need no masking of r: all their features are 0, they add nothing to S)
#pragma unroll
    for (int rb = 0; rb < KB; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) rw[16 * rb + 4 * r] = acc[rb][r] * inv;
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---- S += R . Phi: four rows per MFMA step -------------------------------------------------------
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      const double* zt = zk + (size_t)(4 * t4) * ZS;
      double bv[NCB], av[KB];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) bv[cb] = zt[spa[cb]] * zt[spb[cb]];
#pragma unroll
      for (int rb = 0; rb < KB; ++rb) av[rb] = rk[4 * t4 * RS2 + 16 * rb];
#pragma unroll
      for (int rb = 0; rb < KB; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          sacc[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[rb], bv[cb], sacc[rb][cb], 0, 0, 0);
    }
    if (++since_flush == 64) {         // K^64 <= 64^64 = 2^384 stays inside the float64 range
      sc_lse += log(sc_prod);
      sc_prod = 1.0;
      since_flush = 0;
    }
  }

  // ---- per-workgroup partial block: the eight waves' blocks added in wave order, one accumulator quad at a time ----
  const int FT = a.F16_total;
  const size_t pstride = (size_t)a.K16 * 16 * FT + 4;
  double* P = a.partials + (size_t)blockIdx.x * pstride;
  double* red = Th;                                    // [8 waves][4][64] per quad: 16 KB of the (now idle) operand image
#pragma unroll
  for (int rb = 0; rb < KB; ++rb) {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      wg_sync();
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * 4 + r) * 64 + lane] = sacc[rb][cb][r];
      wg_sync();
      if (wave < 4) {                                  // wave r adds register r of the eight waves
        double s2 = red[(0 * 4 + wave) * 64 + lane];
#pragma unroll
        for (int w = 1; w < kRowWaveWG / 64; ++w) s2 += red[(w * 4 + wave) * 64 + lane];
        const int i = q + 4 * wave;                    // S-row of register `wave` in lane (q, j)
        const int k = (i & 3) * V + 4 * rb + (i >> 2);
        if (k < a.K16 * 16) P[(size_t)k * FT + 16 * cb + j] = k < K ? s2 : 0.0;
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
    for (int w = 0; w < kRowWaveWG / 64; ++w) s2 += sred[w];
    Ps[0] = s2; Ps[1] = 0.0; Ps[2] = 0.0; Ps[3] = 0.0;
  }
}

size_t vi_rowwave_lds_bytes(int KB, int NS, int ZS) {
  return sizeof(double) * ((size_t)(NS * KB + 4) * 64 + kExpTab + (size_t)(kRowWaveWG / 64) * 16 * (ZS + 16 * KB + 8) + 8);
}

// K <= 64, Dz <= 9 (F16 <= 64), from the same K as the label route
bool vi_rowwave_covers(int K, int F16, int ZS) {
  static const bool on = [] { const char* e = getenv("MIMO_ROWWAVE_VI"); return !e || atoi(e) != 0; }();   // tuning knob
  if (!on || K < rowwave_min_k() || K > 64 || F16 > 64) return false;
  if (K > 32 && F16 > 48) return false;       // Dz = 9 with four row blocks: 128 + 32 accumulator registers spill (136 B)
  // measured against the tile kernels (tools/vi_route_time.py, N = 1e7, pass kernels): K <= 32 wins (D = 8: K = 32 1.87 -> 1.37 ms,
  // K = 24 1.80 -> 1.38, K = 16 1.39 -> 1.33; D = 5, K = 32 1.39 -> 1.03; D = 3, K = 24 1.04 -> 0.75) except Dz = 9 at K <= 16
  // (1.47 -> 1.58); 33 <= K <= 48 pays for 64 component slots (D = 7, K = 33: 2.07 -> 2.33); 49 <= K <= 64 wins a little at
  // Dz = 5 .. 8 (D = 8: 2.53 -> 2.41) and loses at Dz <= 4 (D = 1: 1.18 -> 1.32)
  if (K <= 32) { if (F16 == 64 && K <= 16) return false; }
  else if (K < 49 || F16 < 32) return false;
  return vi_rowwave_lds_bytes(K <= 32 ? 2 : 4, F16 / 4, ZS) <= 160 * 1024;
}

hipError_t launch_vi_rowwave(const KernelArgs& a, int grid, hipStream_t stream) {
  typedef void (*fn_t)(const KernelArgs);
  static const fn_t t2[4] = {vi_rowwave_kernel<2, 1>, vi_rowwave_kernel<2, 2>, vi_rowwave_kernel<2, 3>, vi_rowwave_kernel<2, 4>};
  static const fn_t t4[4] = {vi_rowwave_kernel<4, 1>, vi_rowwave_kernel<4, 2>, vi_rowwave_kernel<4, 3>, vi_rowwave_kernel<4, 4>};
  const int kb = a.K <= 32 ? 2 : 4, ns4 = a.F16 / 16;
  if (ns4 < 1 || ns4 > 4 || a.K > 64) return hipErrorInvalidValue;
  fn_t fn = kb == 2 ? t2[ns4 - 1] : t4[ns4 - 1];
  const size_t lds = vi_rowwave_lds_bytes(kb, a.F16 / 4, a.ZS);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(kRowWaveWG), lds, stream, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Statistics of hard labels, K <= 256, Dz <= 9, full feature map.  Workgroup = 256 threads; with Kp = the power of two
// >= K, thread t works for component t % Kp as part t / Kp of P = 256 / Kp: it takes every P-th row of that component's
// list (K > 128: one thread per component; K = 32: eight threads share a component, so all four waves accumulate).
// Its F accumulators (n_k, sum z, upper triangle of sum z z') live in registers across all tiles; the parts of a
// component are added in part order at the end (fixed association).
// Per tile of kLsTile rows:  z tile + labels -> LDS;  bitmap[k] |= 1 << row (integer atomics: the result does not
// depend on their order);  stable position of every row inside its component's list = popcount of the lower bits;
// thread k adds its rows in ascending row order.  Partial block per workgroup in the tile kernels' layout.
// ------------------------------------------------------------------------------------------
constexpr int kLsTile = 512;
constexpr int kLsWideTile = 256;     // tile of the Dz > 10 variants (their z tile is wider)

// FS: feature set — 0: full map (n, sum z, upper triangle of sum z z'), 1: diagonal structure (sum z_a^2, sum z_a, n in the
// order of diag_feat_index), 2: linear structure (sum z_a, n): the reduced maps of mimo_set_structure, Dz <= 16.
template <int DZ, int FS = 0>
__global__ __launch_bounds__(kWG, 2) void label_stats_kernel(const KernelArgs a) {
  constexpr int F = FS == 0 ? (DZ + 1) * (DZ + 2) / 2 : FS == 1 ? 2 * DZ + 1 : DZ + 1;
  constexpr int ZS = DZ <= 2 ? 2 : DZ <= 6 ? 6 : DZ <= 10 ? 10 : DZ <= 14 ? 14 : 18;   // 16-byte aligned rows, odd stride in 16-byte units (random rows: no systematic bank conflicts)
  constexpr int T = DZ <= 10 ? kLsTile : kLsWideTile, NW = T / 32;           // rows per tile, bitmap words per component
  constexpr int RPT = T / kWG;                       // rows per thread and tile
  constexpr int ZPT = (T * DZ + kWG - 1) / kWG;     // z elements per thread
  __shared__ __align__(16) double Zt[T * ZS];
  __shared__ __align__(16) uint32_t bitmap[kWG * NW];   // [k][word] — k-major so that thread k reads 16 consecutive words
  __shared__ uint16_t list[T];
  __shared__ int start[kWG + 1];
  __shared__ int cnts[kWG];
  __shared__ int wsum[4];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K;
  const int64_t N = a.N;
  const int64_t ntiles = (N + T - 1) / T;
  int Kp = 1;
  while (Kp < K) Kp <<= 1;
  const int P = kWG / Kp, myk = tid & (Kp - 1), mypart = tid / Kp;      // (Kp <= 256: K <= 256)

  double acc[F];
#pragma unroll
  for (int f = 0; f < F; ++f) acc[f] = 0.0;

  double zr[ZPT];
  int lab[2] = {-1, -1};
  auto load_tile = [&](int64_t t) {
    const int64_t base = t * T * DZ, total = N * DZ;
#pragma unroll
    for (int i = 0; i < ZPT; ++i) {
      const int64_t g = base + tid + (int64_t)kWG * i;
      zr[i] = (tid + kWG * i < T * DZ && g < total) ? a.Z[g] : 0.0;
    }
#pragma unroll
    for (int h = 0; h < RPT; ++h) {
      const int64_t n = t * T + tid + kWG * h;
      const int l = n < N ? a.labels[n] : -1;
      lab[h] = l < K ? l : -1;            // a label outside [0, K) (a caller's vector) is skipped, never an index
    }
  };
  if (blockIdx.x < ntiles) load_tile(blockIdx.x);
  LS_STAMP_INIT

  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    wg_sync();                        // the previous tile's readers are done
    LS_STAMP(0)
#pragma unroll
    for (int i = 0; i < ZPT; ++i) {
      const int e = tid + kWG * i;
      if (e < T * DZ) { const int r = e / DZ; Zt[r * ZS + (e - r * DZ)] = zr[i]; }
    }
    {
      uint4* bm = reinterpret_cast<uint4*>(bitmap + tid * NW);
#pragma unroll
      for (int w = 0; w < NW / 4; ++w) bm[w] = uint4{0u, 0u, 0u, 0u};
    }
    const int l0 = lab[0], l1 = lab[1];
    if (t + gridDim.x < ntiles) load_tile(t + gridDim.x);
    LS_STAMP(1)
    wg_sync();
    LS_STAMP(2)
    if (l0 >= 0) atomicOr(&bitmap[l0 * NW + (tid >> 5)], 1u << (tid & 31));
    if (RPT > 1 && l1 >= 0) atomicOr(&bitmap[l1 * NW + ((tid + kWG) >> 5)], 1u << (tid & 31));
    wg_sync();
    LS_STAMP(3)
    // rows of component tid, and the exclusive prefix over the components (where its list starts)
    int cntk = 0;
    {
      const uint4* bm = reinterpret_cast<const uint4*>(bitmap + tid * NW);
#pragma unroll
      for (int w = 0; w < NW / 4; ++w) {
        const uint4 v = bm[w];
        cntk += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
      }
    }
    int incl = cntk;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      const int v = __shfl_up(incl, s);
      if (lane >= s) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    wg_sync();
    int off = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) off += w < wave ? wsum[w] : 0;
    start[tid] = off + incl - cntk;
    cnts[tid] = cntk;
    wg_sync();
    LS_STAMP(4)
    // stable position of each row in its component's list
    auto place = [&](int l, int row) {
      if (l < 0) return;
      const uint32_t* bm = bitmap + l * NW;
      const int wq = row >> 5;
      int rank = __popc(bm[wq] & ((1u << (row & 31)) - 1u));
      for (int w = 0; w < wq; ++w) rank += __popc(bm[w]);
      list[start[l] + rank] = (uint16_t)row;
    };
    place(l0, tid);
    if (RPT > 1) place(l1, tid + kWG);
    wg_sync();
    LS_STAMP(5)
    // thread (component myk, part mypart): every P-th row of the component's list, ascending
    const int st = start[myk], cmine = cnts[myk];
    for (int p = mypart; p < cmine; p += P) {
      const int row = list[st + p];
      const double* zp = Zt + row * ZS;
      double z[DZ];
#pragma unroll
      for (int d = 0; d < DZ; ++d) z[d] = zp[d];
      if constexpr (FS == 0) {
        int f = 0;
#pragma unroll
        for (int i = 0; i < DZ; ++i) {
#pragma unroll
          for (int jx = i; jx < DZ; ++jx) { acc[f] = fma(z[i], z[jx], acc[f]); ++f; }
          acc[f] += z[i]; ++f;
        }
      } else if constexpr (FS == 1) {
#pragma unroll
        for (int i = 0; i < DZ; ++i) { acc[i] = fma(z[i], z[i], acc[i]); acc[DZ + i] += z[i]; }
      } else {
#pragma unroll
        for (int i = 0; i < DZ; ++i) acc[i] += z[i];
      }
      acc[F - 1] += 1.0;
    }
    LS_STAMP(6)
  }
  LS_STAMP_STORE

  // per-workgroup partial block [16 K16][F16_total] (+ 4 scalars: none from this pass)
  const int FT = a.F16_total;
  const size_t pstride = (size_t)a.K16 * 16 * FT + 4;
  double* P_out = a.partials + (size_t)blockIdx.x * pstride;
  if (P > 1) {
    // add the parts of every component in part order, eight features at a time through LDS
    static_assert(sizeof(double) * T * ZS >= sizeof(double) * kWG * 8 || sizeof(uint32_t) * kWG * NW >= sizeof(double) * kWG * 8,
                  "reduction scratch fits the bitmap or the z tile");
    double* red = sizeof(uint32_t) * kWG * NW >= sizeof(double) * kWG * 8 ? reinterpret_cast<double*>(bitmap) : Zt;   // [kWG][8]
#pragma unroll
    for (int f0 = 0; f0 < F; f0 += 8) {
      wg_sync();
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (f0 + i < F) red[tid * 8 + i] = acc[f0 + i];
      wg_sync();
      if (mypart == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (f0 + i < F) {
            double s = acc[f0 + i];
            for (int q = 1; q < P; ++q) s += red[(q * Kp + myk) * 8 + i];
            acc[f0 + i] = s;
          }
        }
      }
    }
  }
  if (mypart == 0 && myk < a.K16 * 16) {
#pragma unroll
    for (int f = 0; f < F; ++f) P_out[(size_t)myk * FT + f] = myk < K ? acc[f] : 0.0;
  }
  if (P > 1 && Kp < a.K16 * 16) {       // rows of the partial block between Kp and 16 K16 (K = 17 .. 31 -> Kp = 32 covers them; K <= 16 -> Kp = 16 = 16 K16)
    for (int k = Kp + tid; k < a.K16 * 16; k += kWG)
      for (int f = 0; f < F; ++f) P_out[(size_t)k * FT + f] = 0.0;
  }
  if (tid == 0 && a.write_scalars) {
    double* Ps = P_out + (size_t)a.K16 * 16 * FT;
    Ps[0] = 0.0; Ps[1] = 0.0; Ps[2] = 0.0; Ps[3] = 0.0;
  }
}

// ------------------------------------------------------------------------------------------
// The same pass with the 256 threads of a workgroup assigned to components IN PROPORTION TO THEIR ROWS (K >= 17, Dz <= 9,
// N >= 2^17).  label_stats_kernel gives every component the same number of threads; a DP-GMM sweep at Kmax = 256 keeps its
// rows on a few dozen components, so one lane in eight works while the others wait for it (N = 1e7, Dz = 8, K = 256: 378 us with
// the rows on 32 components against 239 us for uniformly drawn labels; phase stamps of tools/stamps_label_stats.py: list
// placement 22 %, accumulation 22 %, prefix scan 13 % of the wave time, the rest waiting for the tile).  Here:
//   label_hist_kernel    counts the labels of the whole launch (integer atomics: order-free),
//   label_slots_kernel   hands out the 256 slots: one per NON-EMPTY component, the rest in proportion to the counts,
//   label_stats_slots_kernel  slot (component k, part p of n_k) takes the rows of rank p, p + n_k, .. of ITS component's
//                        ascending list of the tile; the parts of a component are added in part order at the end.
// The result is a function of the label vector alone (the slot table is built from it): run-to-run bit-identical.
// Also new against label_stats_kernel: a row's place in the list is one lookup (per-component prefix of the bitmap's word
// popcounts) instead of a loop over the words below it; rows on an odd stride read with 8-byte loads; bitmap rows padded off
// the 64-byte stride that put every component's words into the same banks.
// What the slot table cannot fix: the labels of a real C3 sweep (tools/c3_label_stats.py) sit on 34 components above 1 % — and on 205
// more with a handful of rows each, so 239 of the 256 slots are taken before any helper is handed out (369 us with this kernel, 372 us
// with label_stats_kernel, N = 1e7).  Taking the slots away from components under 1 row in 1024 and adding their rare rows straight
// into the partial block in global memory was tried: a row costs a dependent L2 round trip per feature there, and the busiest owner
// thread holds every tile's barrier (4.4 ms).  Those rows need accumulators next to the CU — LDS has no room for 205 x 48 doubles
// next to the tile — or a sorted second pass; neither is built.
// Tried and dropped on the way (tools/label_stats_time.py): a kernel that walked the set bits of a component's bitmap
// directly (no prefix scan, no list): its divergent bit loop cost ~1000 cycles per iteration whatever the body (152 against
// 87 us, Dz = 8, K = 256, N = 2e6); and 512-thread workgroups over 1024-row tiles with 256 helper slots: indifferent to the
// skew (309 us either way) but one workgroup per CU, whose five serial phases nothing overlaps (239 us before, uniform labels).
// ------------------------------------------------------------------------------------------
#ifndef MIMO_LS_PREFETCH2
#define MIMO_LS_PREFETCH2 0          // 1: two tiles in flight ahead of the one in LDS — measured: no change (D=8 K=256 N=1e7 246.5 -> 244.7 us; D=9 spills: 271 -> 310 us)
#endif
#ifndef MIMO_LS_ROW_PIPELINE
#define MIMO_LS_ROW_PIPELINE 0       // 1: next row's index and values fetched under this row's products — measured: slower (246.5 -> 267.1 us)
#endif
#ifndef MIMO_LS_PAIRS
#define MIMO_LS_PAIRS 1              // even Dz: rows travel HBM -> registers -> LDS 16 bytes at a time (half the load / store instructions of the staging phase)
#endif
constexpr int kLsSlots = kWG;                  // aux layout (uint32): hist[256] | nparts[256] | first slot[256] | slot table[256]
constexpr int kLsAuxWords = 256 * 4;

__global__ __launch_bounds__(kWG) void label_hist_kernel(const int32_t* __restrict__ labels, int64_t N, int K, uint32_t* __restrict__ aux) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0u;
  wg_sync();
  for (int64_t n = (int64_t)blockIdx.x * kWG + threadIdx.x; n < N; n += (int64_t)gridDim.x * kWG) {
    const int l = labels[n];
    if (l >= 0 && l < K) atomicAdd(&h[l], 1u);
  }
  wg_sync();
  if (h[threadIdx.x]) atomicAdd(&aux[threadIdx.x], h[threadIdx.x]);
}

__global__ __launch_bounds__(kWG) void label_slots_kernel(uint32_t* __restrict__ aux, int K) {
  __shared__ uint32_t sc[kWG];
  const int k = threadIdx.x;
  const uint32_t ck = k < K ? aux[k] : 0u;
  sc[k] = ck;
  wg_sync();
  unsigned long long tot = 0ull;
  uint32_t ne = 0u;
  for (int i = 0; i < kWG; ++i) { tot += sc[i]; ne += sc[i] ? 1u : 0u; }
  // one slot per non-empty component + floor(spare count_k / total) of the spare ones (their sum cannot exceed the spare)
  const uint32_t nk = ck ? 1u + (uint32_t)(((unsigned long long)(kLsSlots - ne) * ck) / tot) : 0u;
  wg_sync();
  sc[k] = nk;
  wg_sync();
  uint32_t base = 0u, used = 0u;
  for (int i = 0; i < kWG; ++i) { if (i < k) base += sc[i]; used += sc[i]; }
  aux[256 + k] = nk;
  aux[512 + k] = base;
  wg_sync();
  for (uint32_t j = 0; j < nk; ++j) aux[768 + base + j] = (uint32_t)k | (j << 16);
  if ((uint32_t)k >= used) aux[768 + k] = 0xffffffffu;                 // slots nobody got
}

constexpr int ls_feat(int DZ, int FS) { return FS == 0 ? (DZ + 1) * (DZ + 2) / 2 : FS == 1 ? 2 * DZ + 1 : DZ + 1; }

template <int DZ, int FS = 0>
__global__ __launch_bounds__(kWG, (DZ <= 2 ? 3 : 2)) void label_stats_slots_kernel(const KernelArgs a) {
  constexpr int F = ls_feat(DZ, FS);
#ifdef MIMO_LS_ODD_STRIDE
  constexpr int ZS = DZ | 1;
#else
  constexpr int ZS = DZ <= 2 ? 2 : DZ <= 6 ? 6 : DZ <= 10 ? 10 : DZ <= 14 ? 14 : 18;   // 16-byte aligned rows, odd stride in 16-byte units
#endif
  constexpr int T = kLsTile, NW = T / 32;                  // 512 rows, 16 bitmap words per component
  constexpr int BS = NW + 4, PS = NW + 8;                  // padded row strides of the bitmap (words: 80 bytes) and of its prefix table (u16: 48 bytes)
  constexpr int RPT = T / kWG;                             // 2 rows per thread and tile
  constexpr int ZPT = (T * DZ + kWG - 1) / kWG;
  constexpr int PFD = MIMO_LS_PREFETCH2 ? 2 : 1;            // tiles in flight ahead of the one in LDS
  constexpr bool PAIRS = MIMO_LS_PAIRS && DZ % 2 == 0 && ZS % 2 == 0 && ZPT % 2 == 0;
  __shared__ __align__(16) double Zt[T * ZS > kWG * 8 ? T * ZS : kWG * 8];           // (the epilogue's red[256][8] aliases it)
  __shared__ __align__(16) uint32_t bitmap[kWG * BS];
  __shared__ __align__(16) uint16_t wpre[kWG * PS];                                  // set bits below word w of component k
  __shared__ uint16_t list[T];
  __shared__ int start[kWG];
  __shared__ int cnts[kWG];
  __shared__ int wsum[4];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K;
  const int64_t N = a.N;
  const int64_t ntiles = (N + T - 1) / T;
  const uint32_t* aux = a.aux;
  const uint32_t ent = aux[768 + tid];
  const bool live = ent != 0xffffffffu;
  const int myk = live ? (int)(ent & 0xffffu) : 0, mypart = live ? (int)(ent >> 16) : 0;
  const int nparts = live ? (int)aux[256 + myk] : 1;

  double acc[F];
#pragma unroll
  for (int f = 0; f < F; ++f) acc[f] = 0.0;

  // TWO tiles in flight ahead of the one in LDS (round 4; one before: the stamps showed 38 % of a wave's time in the phase that waits
  // for the prefetched rows — 16 KB per workgroup, 32 KB per CU in flight do not cover the HBM latency at 8 TB/s / 256 CUs)
  double zr[ZPT], zr2[ZPT];
  int lab[RPT], lab2[RPT];
  auto load_tile = [&](int64_t t, double (&zd)[ZPT], int (&ld)[RPT]) {
    const int64_t base = t * T * DZ, total = N * DZ;
    if constexpr (PAIRS) {             // element pair e = 2 (tid + 256 i): both in one row (Dz even), 16-byte aligned in HBM and in LDS
      typedef double d2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int i = 0; i < ZPT / 2; ++i) {
        const int e = 2 * (tid + kWG * i);
        const int64_t g = base + e;
        d2 v = d2{0.0, 0.0};
        if (e < T * DZ && g < total) v = *reinterpret_cast<const d2*>(a.Z + g);      // (g even, total even: the pair is inside the data)
        zd[2 * i] = v.x; zd[2 * i + 1] = v.y;
      }
    } else {
#pragma unroll
      for (int i = 0; i < ZPT; ++i) {
        const int64_t g = base + tid + (int64_t)kWG * i;
        zd[i] = (tid + kWG * i < T * DZ && g < total) ? a.Z[g] : 0.0;
      }
    }
#pragma unroll
    for (int h = 0; h < RPT; ++h) {
      const int64_t n = t * T + tid + kWG * h;
      const int l = n < N ? a.labels[n] : -1;
      ld[h] = l < K ? l : -1;            // a label outside [0, K) (a caller's vector) is skipped, never an index
    }
  };
  if (blockIdx.x < ntiles) load_tile(blockIdx.x, zr, lab);
  if (PFD == 2 && (int64_t)blockIdx.x + gridDim.x < ntiles) load_tile((int64_t)blockIdx.x + gridDim.x, zr2, lab2);
  LS_STAMP_INIT

  auto process = [&](int64_t t, double (&zb)[ZPT], int (&lb)[RPT]) {      // tile t from the register set (zb, lb), which is then refilled two tiles ahead
    wg_sync();                        // the previous tile's readers are done
    LS_STAMP(0)
    if constexpr (PAIRS) {
      typedef double d2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int i = 0; i < ZPT / 2; ++i) {
        const int e = 2 * (tid + kWG * i);
        if (e < T * DZ) { const int r = e / DZ; *reinterpret_cast<d2*>(Zt + r * ZS + (e - r * DZ)) = d2{zb[2 * i], zb[2 * i + 1]}; }
      }
    } else {
#pragma unroll
      for (int i = 0; i < ZPT; ++i) {
        const int e = tid + kWG * i;
        if (e < T * DZ) { const int r = e / DZ; Zt[r * ZS + (e - r * DZ)] = zb[i]; }
      }
    }
    {
      uint4* bm = reinterpret_cast<uint4*>(bitmap + tid * BS);
#pragma unroll
      for (int w = 0; w < NW / 4; ++w) bm[w] = uint4{0u, 0u, 0u, 0u};
    }
    int l01[RPT];
#pragma unroll
    for (int h = 0; h < RPT; ++h) l01[h] = lb[h];
#ifndef MIMO_LS_WHATIF_NOLOAD          // (diagnostic what-if: every tile re-uses the first tile's rows — no HBM traffic)
    if (t + PFD * (int64_t)gridDim.x < ntiles) load_tile(t + PFD * (int64_t)gridDim.x, zb, lb);
#endif
    LS_STAMP(1)
    wg_sync();
    LS_STAMP(2)
#pragma unroll
    for (int h = 0; h < RPT; ++h)
      if (l01[h] >= 0) atomicOr(&bitmap[l01[h] * BS + ((tid + kWG * h) >> 5)], 1u << (tid & 31));
    wg_sync();
    LS_STAMP(3)
    // rows of component tid, the prefix of its bitmap words, and the exclusive prefix over the components
    int cntk = 0;
    {
      const uint4* bm = reinterpret_cast<const uint4*>(bitmap + tid * BS);
      uint4* wp = reinterpret_cast<uint4*>(wpre + tid * PS);
#pragma unroll
      for (int w8 = 0; w8 < NW / 8; ++w8) {              // eight words in, eight 16-bit prefixes out
        const uint4 v0 = bm[2 * w8], v1 = bm[2 * w8 + 1];
        const uint32_t p0 = cntk;           cntk += __popc(v0.x);
        const uint32_t p1 = cntk;           cntk += __popc(v0.y);
        const uint32_t p2 = cntk;           cntk += __popc(v0.z);
        const uint32_t p3 = cntk;           cntk += __popc(v0.w);
        const uint32_t p4 = cntk;           cntk += __popc(v1.x);
        const uint32_t p5 = cntk;           cntk += __popc(v1.y);
        const uint32_t p6 = cntk;           cntk += __popc(v1.z);
        const uint32_t p7 = cntk;           cntk += __popc(v1.w);
        wp[w8] = uint4{p0 | (p1 << 16), p2 | (p3 << 16), p4 | (p5 << 16), p6 | (p7 << 16)};
      }
    }
    int incl = cntk;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
      const int v = __shfl_up(incl, sft);
      if (lane >= sft) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    wg_sync();
    int off = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) off += w < wave ? wsum[w] : 0;
    start[tid] = off + incl - cntk;
    cnts[tid] = cntk;
    wg_sync();
    LS_STAMP(4)
    // stable position of each row in its component's list: one prefix lookup + one popcount
#pragma unroll
    for (int h = 0; h < RPT; ++h) {
      const int l = l01[h], row = tid + kWG * h;
      if (l >= 0) {
        const int wq = row >> 5;
        const int rank = (int)wpre[l * PS + wq] + __popc(bitmap[l * BS + wq] & ((1u << (row & 31)) - 1u));
        list[start[l] + rank] = (uint16_t)row;
      }
    }
    wg_sync();
    LS_STAMP(5)
    // slot (component myk, part mypart of nparts): every nparts-th row of the component's list, ascending
    if (live) {
      const int st = start[myk], cmine = cnts[myk];
#if MIMO_LS_ROW_PIPELINE
      // the next row's index and values travel LDS -> registers under this row's products (list entry -> row is a dependent pair of
      // LDS round trips: exposed, it costs more than the F products of a row)
      int p = mypart;
      double zn[DZ];
      int rown = p + nparts < cmine ? (int)list[st + p + nparts] : 0;
      if (p < cmine) {
        const double* zp0 = Zt + (int)list[st + p] * ZS;
#pragma unroll
        for (int d = 0; d < DZ; ++d) zn[d] = zp0[d];
      }
      for (; p < cmine; p += nparts) {
        double z[DZ];
#pragma unroll
        for (int d = 0; d < DZ; ++d) z[d] = zn[d];
        if (p + nparts < cmine) {
          const double* zp1 = Zt + rown * ZS;
#pragma unroll
          for (int d = 0; d < DZ; ++d) zn[d] = zp1[d];
          rown = p + 2 * nparts < cmine ? (int)list[st + p + 2 * nparts] : 0;
        }
#else
      for (int p = mypart; p < cmine; p += nparts) {
        const int row = list[st + p];
        const double* zp = Zt + row * ZS;
        double z[DZ];
#pragma unroll
        for (int d = 0; d < DZ; ++d) z[d] = zp[d];
#endif
        if constexpr (FS == 0) {
          int f = 0;
#pragma unroll
          for (int i = 0; i < DZ; ++i) {
#pragma unroll
            for (int jx = i; jx < DZ; ++jx) { acc[f] = fma(z[i], z[jx], acc[f]); ++f; }
            acc[f] += z[i]; ++f;
          }
        } else if constexpr (FS == 1) {
#pragma unroll
          for (int i = 0; i < DZ; ++i) { acc[i] = fma(z[i], z[i], acc[i]); acc[DZ + i] += z[i]; }
        } else {
#pragma unroll
          for (int i = 0; i < DZ; ++i) acc[i] += z[i];
        }
        acc[F - 1] += 1.0;
      }
    }
    LS_STAMP(6)
  };
  if constexpr (PFD == 2) {
    for (int64_t t = blockIdx.x; t < ntiles; t += 2 * (int64_t)gridDim.x) {     // two register sets in turn: no copies, every load two tiles ahead of its use
      process(t, zr, lab);
      if (t + gridDim.x < ntiles) process(t + gridDim.x, zr2, lab2);
    }
  } else {
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) process(t, zr, lab);
  }
  LS_STAMP_STORE

  // per-workgroup partial block [16 K16][F16_total]: the slots of a component are neighbours; part 0 adds them in part order, eight
  // features at a time, and writes the component's row; a component without rows in the whole launch has no slot: zeros
  const int FT = a.F16_total;
  const size_t pstride = (size_t)a.K16 * 16 * FT + 4;
  double* P_out = a.partials + (size_t)blockIdx.x * pstride;
  double* red = Zt;                                      // [256][8]
  const bool owner = live && mypart == 0;
#pragma unroll
  for (int f0 = 0; f0 < F; f0 += 8) {
    wg_sync();
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (f0 + i < F) red[tid * 8 + i] = acc[f0 + i];
    wg_sync();
    if (owner) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (f0 + i < F) {
          double s2 = acc[f0 + i];
          for (int q = 1; q < nparts; ++q) s2 += red[(tid + q) * 8 + i];
          acc[f0 + i] = s2;
        }
      }
    }
  }
  if (owner) {
#pragma unroll
    for (int f = 0; f < F; ++f) P_out[(size_t)myk * FT + f] = acc[f];
  }
  if (tid < a.K16 * 16 && (tid >= K || aux[256 + tid] == 0u)) {
    for (int f = 0; f < F; ++f) P_out[(size_t)tid * FT + f] = 0.0;
  }
  if (tid == 0 && a.write_scalars) {
    double* Ps = P_out + (size_t)a.K16 * 16 * FT;
    Ps[0] = 0.0; Ps[1] = 0.0; Ps[2] = 0.0; Ps[3] = 0.0;
  }
}

// ------------------------------------------------------------------------------------------
// The same pass for Dz = 10 .. 16 (F = 66 .. 153 features: too many accumulators for one thread).  The 256 / Kp threads of
// a component split the FEATURES first — thread slice s takes the rows i = s, s + FP, ... of the upper triangle
// (sum z_i z_j for j >= i, and sum z_i; slice 0 also the count) — and, if threads are left (Kp < 256 / FP), the rows of
// the component's list as in label_stats_kernel.  FP = 4 for K <= 64, 2 for K <= 128 (Dz <= 12).  Tiles of 256 rows.
// ------------------------------------------------------------------------------------------
constexpr int slice_count(int DZ, int FP, int S) {
  int n = S == 0 ? 1 : 0;
  for (int i = S; i < DZ; i += FP) n += DZ - i + 1;
  return n;
}

template <int DZ, int FP, int S, int MAXA>
__device__ __forceinline__ void slice_accumulate(double (&acc)[MAXA], const double (&z)[DZ]) {
  int a = 0;
#pragma unroll
  for (int i = S; i < DZ; i += FP) {
#pragma unroll
    for (int j = i; j < DZ; ++j) { acc[a] = fma(z[i], z[j], acc[a]); ++a; }
    acc[a] += z[i]; ++a;
  }
  if constexpr (S == 0) acc[a] += 1.0;
}

template <int DZ, int FP, int S, int MAXA>
__device__ __forceinline__ void slice_store(const double (&acc)[MAXA], double* __restrict__ Pk) {
  constexpr int F = (DZ + 1) * (DZ + 2) / 2;
  int a = 0;
#pragma unroll
  for (int i = S; i < DZ; i += FP) {
    const int f0 = i * (DZ + 1) - i * (i - 1) / 2;       // feature (i, i); (i, j) follows at f0 + j - i, (i, DZ) at f0 + DZ - i
#pragma unroll
    for (int j = i; j < DZ; ++j) { Pk[f0 + j - i] = acc[a]; ++a; }
    Pk[f0 + DZ - i] = acc[a]; ++a;
  }
  if constexpr (S == 0) Pk[F - 1] = acc[a];
}

template <int DZ, int FP>
__global__ __launch_bounds__(kWG, 2) void label_stats_wide_kernel(const KernelArgs a) {
  constexpr int F = (DZ + 1) * (DZ + 2) / 2;
  constexpr int ZS = DZ <= 10 ? 10 : DZ <= 14 ? 14 : 18;     // 16-byte aligned rows, odd stride in 16-byte units
  constexpr int T = kLsWideTile, NW = T / 32;
  constexpr int ZPT = (T * DZ + kWG - 1) / kWG;
  constexpr int MAXA = slice_count(DZ, FP, 0);                // slice 0 is the largest
  static_assert(FP == 2 || FP == 4, "feature slices per component");
  __shared__ __align__(16) double Zt[T * ZS];
  __shared__ __align__(16) uint32_t bitmap[kWG * NW];
  __shared__ __align__(16) uint16_t list[T];
  __shared__ int start[kWG + 1];
  __shared__ int cnts[kWG];
  __shared__ int wsum[4];
  __shared__ double red[kWG * 8];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the launch takes the components k0 .. k0 + K - 1 of the a.K the labels run over (a window of at most 256 / FP: K > 128 at
  // Dz = 10 .. 16 is two launches of 128 — each reads Z once — instead of four feature slices through label_stats_xwide_kernel)
  const int k0 = a.k0;
  const int K = a.K - k0 < kWG / FP ? a.K - k0 : kWG / FP;
  const int64_t N = a.N;
  const int64_t ntiles = (N + T - 1) / T;
  int Kp = 1;
  while (Kp < K) Kp <<= 1;
  const int P = kWG / Kp, RP = P / FP;
  const int myk = tid & (Kp - 1), part = tid / Kp, fslice = part % FP, rpart = part / FP;

  double acc[MAXA];
#pragma unroll
  for (int i = 0; i < MAXA; ++i) acc[i] = 0.0;

  double zr[ZPT];
  int lab;
  auto load_tile = [&](int64_t t) {
    const int64_t base = t * T * DZ, total = N * DZ;
#pragma unroll
    for (int i = 0; i < ZPT; ++i) {
      const int64_t g = base + tid + (int64_t)kWG * i;
      zr[i] = (tid + kWG * i < T * DZ && g < total) ? a.Z[g] : 0.0;
    }
    const int64_t n = t * T + tid;
    const int l = ((n < N && !a.presort) ? a.labels[n] : -1) - k0;
    lab = (l >= 0 && l < K) ? l : -1;       // outside the window (or outside [0, a.K): a caller's vector): skipped
  };
  if (blockIdx.x < ntiles) load_tile(blockIdx.x);

  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    wg_sync();
#pragma unroll
    for (int i = 0; i < ZPT; ++i) {
      const int e = tid + kWG * i;
      if (e < T * DZ) { const int r = e / DZ; Zt[r * ZS + (e - r * DZ)] = zr[i]; }
    }
    int st, cmine;
    if (a.presort) {                   // (uniform) the tile was ranked once for all the launches: label_tile_sort_kernel
      const uint16_t* sg = a.sort_start + (size_t)t * 257 + (k0);
      const bool mine = myk < K;
      const int s0 = mine ? (int)sg[myk] : 0, s1 = mine ? (int)sg[myk + 1] : 0;
      if (tid < T / 2) reinterpret_cast<uint32_t*>(list)[tid] = reinterpret_cast<const uint32_t*>(a.sort_list + (size_t)t * T)[tid];
      if (t + gridDim.x < ntiles) load_tile(t + gridDim.x);
      st = s0; cmine = s1 - s0;
      wg_sync();
    } else {
      {
        uint4* bm = reinterpret_cast<uint4*>(bitmap + tid * NW);
  #pragma unroll
        for (int w = 0; w < NW / 4; ++w) bm[w] = uint4{0u, 0u, 0u, 0u};
      }
      const int l0 = lab;
      if (t + gridDim.x < ntiles) load_tile(t + gridDim.x);
      wg_sync();
      if (l0 >= 0) atomicOr(&bitmap[l0 * NW + (tid >> 5)], 1u << (tid & 31));
      wg_sync();
      int cntk = 0;
      {
        const uint4* bm = reinterpret_cast<const uint4*>(bitmap + tid * NW);
  #pragma unroll
        for (int w = 0; w < NW / 4; ++w) {
          const uint4 v = bm[w];
          cntk += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
        }
      }
      int incl = cntk;
  #pragma unroll
      for (int s = 1; s < 64; s <<= 1) {
        const int v = __shfl_up(incl, s);
        if (lane >= s) incl += v;
      }
      if (lane == 63) wsum[wave] = incl;
      wg_sync();
      int off = 0;
  #pragma unroll
      for (int w = 0; w < 4; ++w) off += w < wave ? wsum[w] : 0;
      start[tid] = off + incl - cntk;
      cnts[tid] = cntk;
      wg_sync();
      if (l0 >= 0) {
        const uint32_t* bm = bitmap + l0 * NW;
        const int wq = tid >> 5;
        int rank = __popc(bm[wq] & ((1u << (tid & 31)) - 1u));
        for (int w = 0; w < wq; ++w) rank += __popc(bm[w]);
        list[start[l0] + rank] = (uint16_t)tid;
      }
      wg_sync();
      st = start[myk]; cmine = cnts[myk];
    }
    for (int p = rpart; p < cmine; p += RP) {
      const int row = list[st + p];
      const double* zp = Zt + row * ZS;
      double z[DZ];
#pragma unroll
      for (int d = 0; d < DZ; ++d) z[d] = zp[d];
      switch (fslice) {      // (uniform per wave when Kp >= 64)
        case 0: slice_accumulate<DZ, FP, 0, MAXA>(acc, z); break;
        case 1: slice_accumulate<DZ, FP, 1, MAXA>(acc, z); break;
        case 2: if constexpr (FP == 4) slice_accumulate<DZ, FP, 2, MAXA>(acc, z); break;
        default: if constexpr (FP == 4) slice_accumulate<DZ, FP, 3, MAXA>(acc, z); break;
      }
    }
  }

  const int FT = a.F16_total;
  const size_t pstride = (size_t)a.K16 * 16 * FT + 4;
  double* P_out = a.partials + (size_t)blockIdx.x * pstride;
  if (RP > 1) {          // add the row parts of every (component, slice) in part order, eight accumulators at a time
#pragma unroll
    for (int f0 = 0; f0 < MAXA; f0 += 8) {
      wg_sync();
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (f0 + i < MAXA) red[tid * 8 + i] = acc[f0 + i];
      wg_sync();
      if (rpart == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (f0 + i < MAXA) {
            double s = acc[f0 + i];
            for (int qq = 1; qq < RP; ++qq) s += red[((qq * FP + fslice) * Kp + myk) * 8 + i];
            acc[f0 + i] = s;
          }
        }
      }
    }
  }
  // rows of the partial block without a component (the first window's launch)
  if (k0 == 0)
    for (int k = a.K + tid; k < a.K16 * 16; k += kWG)
      for (int f = 0; f < F; ++f) P_out[(size_t)k * FT + f] = 0.0;
  if (rpart == 0 && myk < K) {
    double* Pk = P_out + (size_t)(k0 + myk) * FT;
    switch (fslice) {
      case 0: slice_store<DZ, FP, 0, MAXA>(acc, Pk); break;
      case 1: slice_store<DZ, FP, 1, MAXA>(acc, Pk); break;
      case 2: if constexpr (FP == 4) slice_store<DZ, FP, 2, MAXA>(acc, Pk); break;
      default: if constexpr (FP == 4) slice_store<DZ, FP, 3, MAXA>(acc, Pk); break;
    }
  }
  if (tid == 0 && a.write_scalars && k0 == 0) {
    double* Ps = P_out + (size_t)a.K16 * 16 * FT;
    Ps[0] = 0.0; Ps[1] = 0.0; Ps[2] = 0.0; Ps[3] = 0.0;
  }
}

// ------------------------------------------------------------------------------------------
// The same pass where label_stats_wide_kernel's accumulators no longer fit: Dz = 17 .. 32 (F up to 561), and K > 64
// (K > 128 at Dz <= 12) at Dz = 10 .. 16.  The upper triangle is cut into FPT interleaved slices (FPT = 4 up to Dz = 16,
// 8 up to Dz = 29, 16 above: at most 76 accumulators per thread next to the 16 registers that carry the next 128-row tile), a launch takes the min(FPT, 256 / Kp) slices [s0, s0 + FPL) that its
// 256 / Kp threads per component can hold, and the host launches FPT / FPL times into the same partial block (each launch
// reads Z once: 8 N Dz bytes — Dz = 32, K = 128: 8 launches, Dz = 28: 4 — against 6 one-hot MFMA launches of the tile kernel before).  The row of a member is read from the LDS tile per product instead of being copied to registers first
// (z_i once per triangle row, z_j per product: 8 F / FPT bytes of LDS reads per thread and member).
// ------------------------------------------------------------------------------------------
template <int DZ, int FPT, int S, int MAXA>
__device__ __forceinline__ void slice_accumulate_lds(double (&acc)[MAXA], const double* __restrict__ zp) {
  int a = 0;
#pragma unroll
  for (int i = S; i < DZ; i += FPT) {
    const double zi = zp[i];
#pragma unroll
    for (int j = i; j < DZ; ++j) { acc[a] = fma(zi, zp[j], acc[a]); ++a; }
    acc[a] += zi; ++a;
  }
  if constexpr (S == 0) acc[a] += 1.0;
}
// Feature slices FPT and rows per tile of the sliced label statistics.  4 slices x 256 rows up to Dz = 16.  Above: 8 slices
// need the registers of half a prefetched tile, i.e. 128-row tiles (Dz = 24, K = 100: 1.25 -> 0.75 ms against 16 x 256) — but
// half the rows per tile also halve the members per component, and a wave runs as long as its busiest lane: at Dz = 30 .. 32
// (85 accumulators: a small spill on top) that costs more than it gives from K = 65 on (Dz = 32, K = 128: 1.87 against 1.51 ms;
// K = 32: 0.70 against 0.84), so those shapes keep 16 x 256.
constexpr int xwide_tile(int FPT) { return FPT == 8 ? kLsWideTile / 2 : kLsWideTile; }
static int xwide_fpt(int D, int K) { return D <= 16 ? 4 : (D <= 29 || K <= 64) ? 8 : 16; }

template <int DZ, int FPT>
__global__ __launch_bounds__(kWG, 2) void label_stats_xwide_kernel(const KernelArgs a) {
  constexpr int F = (DZ + 1) * (DZ + 2) / 2;
  constexpr int ZS = DZ | 1;                                  // odd: members' rows are arbitrary, a column is conflict-free
  constexpr int T = xwide_tile(FPT), NW = T / 32;
  constexpr int ZPT = (T * DZ + kWG - 1) / kWG;
  constexpr int MAXA = slice_count(DZ, FPT, 0);               // slice 0 is the largest
  extern __shared__ __align__(16) unsigned char smem_ls[];
  double* Zt = reinterpret_cast<double*>(smem_ls);                                   // [T][ZS]; the epilogue's red[kWG * 8] aliases it
  uint32_t* bitmap = reinterpret_cast<uint32_t*>(Zt + (T * ZS > kWG * 8 ? T * ZS : kWG * 8));   // [kWG][NW]
  int* start = reinterpret_cast<int*>(bitmap + kWG * NW);                            // [kWG + 1]
  int* cnts = start + kWG + 1;                                                       // [kWG]
  int* wsum = cnts + kWG;                                                            // [4]
  uint16_t* list = reinterpret_cast<uint16_t*>(wsum + 4);                            // [T]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K;
  const int64_t N = a.N;
  const int64_t ntiles = (N + T - 1) / T;
  int Kp = 1;
  while (Kp < K) Kp <<= 1;                                    // K <= 256 = kWG
  const int P = kWG / Kp;                                     // threads per component
  const int FPL = P < FPT ? P : FPT, RP = P / FPL;            // slices of this launch, row parts per (component, slice)
  const int myk = tid & (Kp - 1), part = tid / Kp, fslice = part % FPL, rpart = part / FPL;
  const int myslice = a.cb0 + fslice;                         // (uniform per wave when Kp >= 64)

  double acc[MAXA];
#pragma unroll
  for (int i = 0; i < MAXA; ++i) acc[i] = 0.0;

  double zr[ZPT];
  int lab;
  auto load_tile = [&](int64_t t) {
    const int64_t base = t * T * DZ, total = N * DZ;
#pragma unroll
    for (int i = 0; i < ZPT; ++i) {
      const int64_t g = base + tid + (int64_t)kWG * i;
      zr[i] = (tid + kWG * i < T * DZ && g < total) ? a.Z[g] : 0.0;
    }
    const int64_t n = t * T + tid;
    const int l = (tid < T && n < N && !a.presort) ? a.labels[n] : -1;
    lab = l < K ? l : -1;
  };
  if (blockIdx.x < ntiles) load_tile(blockIdx.x);

  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    wg_sync();
#pragma unroll
    for (int i = 0; i < ZPT; ++i) {
      const int e = tid + kWG * i;
      if (e < T * DZ) { const int r = e / DZ; Zt[r * ZS + (e - r * DZ)] = zr[i]; }
    }
    int st, cmine;
    if (a.presort) {                   // (uniform) the tile was ranked once for all the launches: label_tile_sort_kernel
      const uint16_t* sg = a.sort_start + (size_t)t * 257;
      const bool mine = myk < K;
      const int s0 = mine ? (int)sg[myk] : 0, s1 = mine ? (int)sg[myk + 1] : 0;
      if (tid < T / 2) reinterpret_cast<uint32_t*>(list)[tid] = reinterpret_cast<const uint32_t*>(a.sort_list + (size_t)t * T)[tid];
      if (t + gridDim.x < ntiles) load_tile(t + gridDim.x);
      st = s0; cmine = s1 - s0;
      wg_sync();
    } else {
      {
        uint4* bm = reinterpret_cast<uint4*>(bitmap + tid * NW);
  #pragma unroll
        for (int w = 0; w < NW / 4; ++w) bm[w] = uint4{0u, 0u, 0u, 0u};
      }
      const int l0 = lab;
      if (t + gridDim.x < ntiles) load_tile(t + gridDim.x);
      wg_sync();
      if (l0 >= 0) atomicOr(&bitmap[l0 * NW + (tid >> 5)], 1u << (tid & 31));
      wg_sync();
      int cntk = 0;
      {
        const uint4* bm = reinterpret_cast<const uint4*>(bitmap + tid * NW);
  #pragma unroll
        for (int w = 0; w < NW / 4; ++w) {
          const uint4 v = bm[w];
          cntk += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
        }
      }
      int incl = cntk;
  #pragma unroll
      for (int s = 1; s < 64; s <<= 1) {
        const int v = __shfl_up(incl, s);
        if (lane >= s) incl += v;
      }
      if (lane == 63) wsum[wave] = incl;
      wg_sync();
      int off = 0;
  #pragma unroll
      for (int w = 0; w < 4; ++w) off += w < wave ? wsum[w] : 0;
      start[tid] = off + incl - cntk;
      cnts[tid] = cntk;
      wg_sync();
      if (l0 >= 0) {
        const uint32_t* bm = bitmap + l0 * NW;
        const int wq = tid >> 5;
        int rank = __popc(bm[wq] & ((1u << (tid & 31)) - 1u));
        for (int w = 0; w < wq; ++w) rank += __popc(bm[w]);
        list[start[l0] + rank] = (uint16_t)tid;
      }
      wg_sync();
      st = start[myk]; cmine = cnts[myk];
    }
    if (myslice < FPT)
      for (int p = rpart; p < cmine; p += RP)
      {
        const double* zp = Zt + (int)list[st + p] * ZS;
        switch (myslice) {      // (explicit cases: a recursive template dispatch left acc[] in scratch)
#define MIMO_SL(S) case S: if constexpr (S < FPT) slice_accumulate_lds<DZ, FPT, S, MAXA>(acc, zp); break;
          MIMO_SL(0) MIMO_SL(1) MIMO_SL(2) MIMO_SL(3) MIMO_SL(4) MIMO_SL(5) MIMO_SL(6) MIMO_SL(7)
          MIMO_SL(8) MIMO_SL(9) MIMO_SL(10) MIMO_SL(11) MIMO_SL(12) MIMO_SL(13) MIMO_SL(14) MIMO_SL(15)
#undef MIMO_SL
          default: break;
        }
      }
  }

  const int FT = a.F16_total;
  const size_t pstride = (size_t)a.K16 * 16 * FT + 4;
  double* P_out = a.partials + (size_t)blockIdx.x * pstride;
  if (RP > 1) {          // add the row parts of every (component, slice) in part order, eight accumulators at a time
    double* red = Zt;
#pragma unroll
    for (int f0 = 0; f0 < MAXA; f0 += 8) {
      wg_sync();
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (f0 + i < MAXA) red[tid * 8 + i] = acc[f0 + i];
      wg_sync();
      if (rpart == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (f0 + i < MAXA) {
            double s = acc[f0 + i];
            for (int qq = 1; qq < RP; ++qq) s += red[((qq * FPL + fslice) * Kp + myk) * 8 + i];
            acc[f0 + i] = s;
          }
        }
      }
    }
  }
  // rows of the partial block without a component (first launch)
  if (a.cb0 == 0)
    for (int k = K + tid; k < a.K16 * 16; k += kWG)
      for (int f = 0; f < F; ++f) P_out[(size_t)k * FT + f] = 0.0;
  if (rpart == 0 && myk < K && myslice < FPT) {
    double* Pk = P_out + (size_t)myk * FT;
    // accumulator a of slice s -> feature index (the order of slice_accumulate_lds: rows i = s, s + FPT, ... of the
    // triangle, each with its DZ - i products and the linear term; slice 0 ends with the count).  Computed at run time:
    // a switch over the slices around unrolled stores kept every accumulator live in all 16 cases (3 KB of scratch).
    auto dest = [&](int aidx) -> int {
      int base = 0;
      for (int i = myslice; i < DZ; i += FPT) {
        const int len = DZ - i + 1;
        if (aidx < base + len) return i * (DZ + 1) - i * (i - 1) / 2 + (aidx - base);
        base += len;
      }
      return (myslice == 0 && aidx == base) ? F - 1 : -1;
    };
#pragma unroll
    for (int ai = 0; ai < MAXA; ++ai) {
      const int d = dest(ai);
      if (d >= 0) Pk[d] = acc[ai];
    }
  }
  if (tid == 0 && a.write_scalars) {
    double* Ps = P_out + (size_t)a.K16 * 16 * FT;
    Ps[0] = 0.0; Ps[1] = 0.0; Ps[2] = 0.0; Ps[3] = 0.0;
  }
}

// ------------------------------------------------------------------------------------------
// The ranking of a tile's labels — bitmap per component, popcount ranks, prefix over the components, the rows in component order —
// is the same in every launch of the sliced / windowed label statistics (2 .. 8 launches at Dz >= 10): it runs ONCE here, and the
// launches read the list (2 bytes per row) and the per-component starts (257 x 2 bytes per tile) instead — two workgroup barriers
// per tile instead of six.  T = the consumer's tile (128 or 256 rows).
// ------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(kWG) void label_tile_sort_kernel(const int32_t* __restrict__ labels, int64_t N, int K,
                                                              uint16_t* __restrict__ list_out, uint16_t* __restrict__ start_out) {
  constexpr int NW = T / 32;
  __shared__ __align__(16) uint32_t bitmap[kWG * NW];
  __shared__ int start[kWG + 1];
  __shared__ int wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t ntiles = (N + T - 1) / T;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t n = t * T + tid;
    int l0 = (tid < T && n < N) ? labels[n] : -1;
    if (l0 >= K) l0 = -1;
    {
      uint4* bm = reinterpret_cast<uint4*>(bitmap + tid * NW);
#pragma unroll
      for (int w = 0; w < NW / 4; ++w) bm[w] = uint4{0u, 0u, 0u, 0u};
    }
    wg_sync();
    if (l0 >= 0) atomicOr(&bitmap[l0 * NW + (tid >> 5)], 1u << (tid & 31));
    wg_sync();
    int cntk = 0;
    {
      const uint4* bm = reinterpret_cast<const uint4*>(bitmap + tid * NW);
#pragma unroll
      for (int w = 0; w < NW / 4; ++w) {
        const uint4 v = bm[w];
        cntk += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
      }
    }
    int incl = cntk;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      const int v = __shfl_up(incl, s);
      if (lane >= s) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    wg_sync();
    int off = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) off += w < wave ? wsum[w] : 0;
    start[tid] = off + incl - cntk;
    if (tid == kWG - 1) start[kWG] = off + incl;
    wg_sync();
    uint16_t* so = start_out + (size_t)t * 257;
    so[tid] = (uint16_t)start[tid];
    if (tid == 0) so[256] = (uint16_t)start[kWG];
    if (l0 >= 0) {
      const uint32_t* bm = bitmap + l0 * NW;
      const int wq = tid >> 5;
      int rank = __popc(bm[wq] & ((1u << (tid & 31)) - 1u));
      for (int w = 0; w < wq; ++w) rank += __popc(bm[w]);
      list_out[(size_t)t * T + start[l0] + rank] = (uint16_t)tid;
    }
    wg_sync();                           // bitmap / start are rewritten by the next tile
  }
}
// rank the tiles of a.labels once for the launches of a multi-launch label-statistics pass (tile = rows per tile of the consumer)
static hipError_t launch_label_tile_sort(const KernelArgs& a, int tile, int grid, hipStream_t stream) {
  if (!a.sort_list || !a.sort_start) return hipErrorInvalidValue;
  if (tile == 128) hipLaunchKernelGGL(label_tile_sort_kernel<128>, dim3(grid), dim3(kWG), 0, stream, a.labels, a.N, a.K, a.sort_list, a.sort_start);
  else hipLaunchKernelGGL(label_tile_sort_kernel<256>, dim3(grid), dim3(kWG), 0, stream, a.labels, a.N, a.K, a.sort_list, a.sort_start);
  return hipGetLastError();
}
static bool label_presort_on() {
  static const bool on = [] { const char* e = getenv("MIMO_LABEL_PRESORT"); return !e || atoi(e) != 0; }();   // tuning knob
  return on;
}

// ------------------------------------------------------------------------------------------
// Label statistics from the ranked tiles in ONE pass over Z whatever Dz is (label_stats_sorted_kernel, Dz = 17 .. 32; the sliced
// kernel above reads Z once per launch, 2 .. 16 launches).  The thread <-> accumulator assignment is turned around: the 256 threads of
// a workgroup share the FEATURES (at most three each: F <= 561) and walk the components one after the other — for component k the
// rows of the workgroup's tile range are looked up in the ranked tiles (label_tile_sort_kernel: list + starts), gathered 32 at a time
// into LDS, and every thread adds its features of those rows; then its sums go into the partial block.  Rows are taken in ascending
// order, ranges in ascending order per workgroup, the blocks are reduced in block order: run-to-run bit-identical.
// Per row: F products, 2 LDS reads each — the LDS pipe bounds the pass (~100 cycles per row and CU at Dz = 32), not HBM.
// ------------------------------------------------------------------------------------------
constexpr int kSortedRange = 80;                 // tiles (of 256 rows) per range at most (ids: 40 KB; N = 1e7 on 512 workgroups: 77)
constexpr int kSortedBatch = 64;                 // rows gathered per step
template <int DZ>
__global__ __launch_bounds__(kWG, 2) void label_stats_sorted_kernel(const KernelArgs a, int R) {
  constexpr int F = (DZ + 1) * (DZ + 2) / 2, NF = (F + kWG - 1) / kWG;
  constexpr int T = kLsWideTile, B = kSortedBatch, ZS = (DZ + 2) | 1;
  constexpr int GPT = (B * DZ + kWG - 1) / kWG;                      // gathered elements per thread and batch
  __shared__ __align__(16) double zbuf[B * ZS];
  __shared__ uint16_t ids[kSortedRange * T];                          // (tile in range) << 8 | row in tile: by component, rows ascending
  __shared__ int kbase[kWG + 1];                                      // first position of every component in ids
  __shared__ int wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K;
  const int64_t N = a.N;
  const int64_t ntiles = (N + T - 1) / T, nranges = (ntiles + R - 1) / R;
  const int FT = a.F16_total;
  const size_t pstride = (size_t)a.K16 * 16 * FT + 4;
  double* P = a.partials + (size_t)blockIdx.x * pstride;

  // this thread's features f = tid + 256 j: the two factors inside a z~ row ([z, 1, 0]); none: the zero slot twice
  int oa[NF], ob[NF];
#pragma unroll
  for (int j = 0; j < NF; ++j) {
    const int f = tid + kWG * j;
    oa[j] = f < F ? a.feat[2 * f] : DZ + 1;
    ob[j] = f < F ? a.feat[2 * f + 1] : DZ + 1;
  }
  // rows / columns of the block no accumulator of this kernel reaches, the scalar slots
  for (int e = tid; e < a.K16 * 16 * FT; e += kWG) {
    const int k = e / FT, f = e - k * FT;
    if (k >= K || f >= F) P[e] = 0.0;
  }
  if (tid == 0 && a.write_scalars) { double* Ps = P + (size_t)a.K16 * 16 * FT; Ps[0] = 0.0; Ps[1] = 0.0; Ps[2] = 0.0; Ps[3] = 0.0; }
  if (tid < B) zbuf[tid * ZS + DZ + 1] = 0.0;          // the zero slot (padding features): written once

  bool first = true;
  for (int64_t rg = blockIdx.x; rg < nranges; rg += gridDim.x) {
    const int64_t t0 = rg * R;
    const int nt = (int)(ntiles - t0 < R ? ntiles - t0 : R);
    // ---- the rows of the range by component (thread k = component k): counts, prefix over the components, ids
    wg_sync();                                           // the previous range's readers of ids / kbase are done
    int cntk = 0;
    if (tid < K) {                                       // = start[last tile][..] differences summed: independent loads, eight tiles in flight
#pragma unroll 8
      for (int t = 0; t < nt; ++t) {
        const uint16_t* sg = a.sort_start + (size_t)(t0 + t) * 257 + tid;
        cntk += (int)sg[1] - (int)sg[0];
      }
    }
    int incl = cntk;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
      const int v = __shfl_up(incl, sft);
      if (lane >= sft) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    wg_sync();
    int off = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) off += w < wave ? wsum[w] : 0;
    int o = off + incl - cntk;
    kbase[tid] = o;
    if (tid == kWG - 1) kbase[kWG] = off + incl;
    if (tid < K)
      for (int tb = 0; tb < nt; tb += 4) {               // four tiles' starts (and first list entries) in flight before they are used
        int s0[4], c[4];
        uint16_t l0[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int t = tb + i < nt ? tb + i : nt - 1;
          const uint16_t* sg = a.sort_start + (size_t)(t0 + t) * 257 + tid;
          s0[i] = sg[0]; c[i] = tb + i < nt ? (int)sg[1] - s0[i] : 0;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int t = tb + i < nt ? tb + i : nt - 1;
          l0[i] = c[i] > 0 ? a.sort_list[(size_t)(t0 + t) * T + s0[i]] : (uint16_t)0;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int t = tb + i;
          if (c[i] > 0) {
            ids[o++] = (uint16_t)((t << 8) | l0[i]);
            const uint16_t* lg = a.sort_list + (size_t)(t0 + t) * T + s0[i];
            for (int m = 1; m < c[i]; ++m) ids[o++] = (uint16_t)((t << 8) | lg[m]);
          }
        }
      }
    wg_sync();
    if (first) {                                         // components without a row in the workgroup's first range: zero rows
      for (int e = tid; e < K * F; e += kWG) {
        const int k = e / F, f = e - k * F;
        if (kbase[k + 1] == kbase[k]) P[(size_t)k * FT + f] = 0.0;
      }
    }
    // ---- stream the rows (component order) in full batches of B; the next batch is in flight (global -> registers) while the
    //      current one is accumulated from LDS; a batch is walked in segments of one component each, whose sums go into the block
    //      when its last row is done
    const int nrows = kbase[kWG];
    double gz[2][GPT];                                   // two batches in flight ahead of the one in LDS
    auto fetch = [&](int pos, double (&g)[GPT]) {
      const int nb = nrows - pos < B ? nrows - pos : B;
#pragma unroll
      for (int i = 0; i < GPT; ++i) {
        const int e = tid + kWG * i, r = e / DZ, col = e - r * DZ;
        double v = 0.0;
        if (e < B * DZ && r < nb) {
          const int id = ids[pos + r];
          const int64_t n = (t0 + (id >> 8)) * T + (id & 255);
          v = a.Z[n * DZ + col];
        }
        g[i] = v;
      }
    };
    if (nrows > 0) fetch(0, gz[0]);
    if (nrows > B) fetch(B, gz[1]);
    double acc[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[j] = 0.0;
    int k = 0;
    auto step = [&](int pos, double (&g)[GPT]) {         // batch at pos: its rows sit in g; g is refilled with the batch two ahead
      const int nb = nrows - pos < B ? nrows - pos : B;
#pragma unroll
      for (int i = 0; i < GPT; ++i) {
        const int e = tid + kWG * i, r = e / DZ, col = e - r * DZ;
        if (e < B * DZ) zbuf[r * ZS + col] = g[i];
      }
      if (tid < B) zbuf[tid * ZS + DZ] = 1.0;
      wg_sync();
      if (pos + 2 * B < nrows) fetch(pos + 2 * B, g);
      int r = 0;
      while (r < nb) {                                   // (uniform control flow: kbase is the same for every thread)
        while (kbase[k + 1] <= pos + r) ++k;             // the component of row pos + r (empty ones are stepped over)
        const int kend = kbase[k + 1] - pos;             // its rows end here (batch-relative)
        const int rend = kend < nb ? kend : nb;
        const double* zr = zbuf + r * ZS;
        // several rows' factors in flight (one row at a time is an LDS round trip per row: 3.6 us per batch of 64 measured)
        // (one feature per thread — Dz <= 21 — only: with two or three the row's own reads overlap, and the unrolled form ran slower: Dz = 32 0.41 -> 0.50 ms)
        constexpr int U = 8;
        if constexpr (NF == 1)
        for (; r + U <= rend; r += U, zr += U * ZS) {
          double fa[U][NF], fb[U][NF];
#pragma unroll
          for (int q = 0; q < U; ++q)
#pragma unroll
            for (int j = 0; j < NF; ++j) { fa[q][j] = zr[q * ZS + oa[j]]; fb[q][j] = zr[q * ZS + ob[j]]; }
#pragma unroll
          for (int q = 0; q < U; ++q)
#pragma unroll
            for (int j = 0; j < NF; ++j) acc[j] = fma(fa[q][j], fb[q][j], acc[j]);
        }
        for (; r < rend; ++r, zr += ZS) {
#pragma unroll
          for (int j = 0; j < NF; ++j) acc[j] = fma(zr[oa[j]], zr[ob[j]], acc[j]);
        }
        if (kend <= nb) {                                // component k is complete: first range writes, later ranges add
#pragma unroll
          for (int j = 0; j < NF; ++j) {
            const int f = tid + kWG * j;
            if (f < F) {
              double* q = P + (size_t)k * FT + f;
              *q = first ? acc[j] : *q + acc[j];
            }
            acc[j] = 0.0;
          }
        }
      }
      wg_sync();
    };
    for (int pos = 0; pos < nrows; pos += 2 * B) {
      step(pos, gz[0]);
      if (pos + B < nrows) step(pos + B, gz[1]);
    }
    first = false;
  }
  if (first) {                                           // a workgroup without a range: an all-zero block
    for (int e = tid; e < K * FT; e += kWG) {
      const int k = e / FT, f = e - k * FT;
      if (f < F) P[(size_t)k * FT + f] = 0.0;
    }
  }
}
// ------------------------------------------------------------------------------------------
// The same pass with the accumulation on the matrix cores (round 4).  label_stats_sorted_kernel gives every thread up to three
// features and reads two factors per product from LDS: 2 F 8 bytes of LDS traffic per row (9 KB at Dz = 32) — the LDS pipe
// bounds it at 1.3 - 1.6 TB/s of Z.  But the statistics of ONE component's rows are a Gram matrix, z~' z~ over those rows
// (z~ = [z, 1]: second moments, sums and the count are its entries), and the rows arrive sorted by component: per group of
// four rows one v_mfma_f64_16x16x4_f64 per 16 x 16 tile of the upper triangle (1 tile up to Dz = 15, 3 up to Dz = 31, 6 at
// Dz = 32), operands straight from the gathered rows in LDS — 1 - 2 reads of 512 bytes per matrix instruction, 1.5 KB per
// row at Dz = 32.  Each wave owns tiles of the triangle (no cross-wave sum); a component's accumulators go to the partial
// block when its last row is done (first range writes, later ranges add), rows of a group that belong to the next component
// are masked to zero.  Ranges, id lists and the gather are those of the kernel above; run-to-run bit-identical.
// ------------------------------------------------------------------------------------------
#ifndef MIMO_GRAM_WHATIF
#define MIMO_GRAM_WHATIF 0           // diagnostic builds: 1 no batches (bookkeeping only), 2 no flushes to the partial block (results are wrong)
#endif
#ifndef MIMO_GRAM_BATCH
#define MIMO_GRAM_BATCH 0            // 0: 64 rows per step (128 measured 4 - 6 % faster at two workgroups per CU, but costs the third and fourth)
#endif
// ranges of at most 40 tiles (20 KB of ids) and 64-row batches: 37 - 45 KB of LDS, four (three at Dz = 32) workgroups per CU — the
// pass has three phases of comparable length that one workgroup runs one after the other (range bookkeeping, gather + staging,
// products: what-if builds, profiles/r04_label_stats_gram.txt), so it takes co-resident workgroups to overlap them
// (measured, N = 1e7, two -> four workgroups per CU: Dz=20 K=64 981 -> 779 us, Dz=24 K=200 1149 -> 975, Dz=28 K=16 1052 -> 812, Dz=31 K=128
//  1190 -> 1068 with three; Dz = 32 — six tiles, 45 KB, 1.2 MB of partial block per workgroup at K = 256 — loses with three
//  (K=128 1672 -> 1900, K=256 1858 -> 2400) and keeps two workgroups over ranges of 80 tiles)
constexpr int gram_range(int DZ) { return DZ <= 31 ? 40 : 80; }
constexpr int gram_wgs_per_cu(int DZ) { return DZ <= 31 ? 4 : 2; }      // (registers: 85 .. 133; four per CU cap them at 128: Dz=31 1437 us with three, 1068 with four)
template <int DZ>
__global__ __launch_bounds__(kWG, gram_wgs_per_cu(DZ)) void label_stats_gram_kernel(const KernelArgs a, int R) {
  constexpr int F = (DZ + 1) * (DZ + 2) / 2;
  constexpr int TT = (DZ + 1 + 15) / 16;                               // 16-wide tiles per side of the Gram matrix of z~
  constexpr int NTL = TT * (TT + 1) / 2;                               // tiles of the upper triangle: 1, 3 or 6
  constexpr int TPW = (NTL + 3) / 4;                                   // tiles per wave: 1, 1 or 2
  constexpr int T = kLsWideTile, B = MIMO_GRAM_BATCH > 0 ? MIMO_GRAM_BATCH : 64;    // rows gathered per step
  constexpr int ZS = 16 * TT + 1;                                      // rows [z, 1, 0 ..] padded to whole tiles; odd stride
  constexpr int GPT = (B * DZ + kWG - 1) / kWG;                        // gathered elements per thread and batch
  __shared__ __align__(16) double zbuf[B * ZS];
  __shared__ uint16_t ids[gram_range(DZ) * T];
  __shared__ int kbase[kWG + 1];
  __shared__ int wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, j = lane & 15;
  const int K = a.K;
  const int64_t N = a.N;
  const int64_t ntiles = (N + T - 1) / T, nranges = (ntiles + R - 1) / R;
  const int FT = a.F16_total;
  const size_t pstride = (size_t)a.K16 * 16 * FT + 4;
  double* P = a.partials + (size_t)blockIdx.x * pstride;

  // this wave's tiles (ti <= tj) of the triangle, in row-major order of the triangle: tile index wave + 4 i
  int ti[TPW], tj[TPW];
  bool has[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int t = wave + 4 * i;
    has[i] = t < NTL;
    int r = 0, t0 = 0;
    while (r + 1 < TT && t0 + (TT - r) <= t) { t0 += TT - r; ++r; }
    ti[i] = r; tj[i] = has[i] ? r + (t - t0) : r;
  }
  for (int e = tid; e < a.K16 * 16 * FT; e += kWG) {                  // rows / columns of the block no accumulator reaches
    const int k = e / FT, f = e - k * FT;
    if (k >= K || f >= F) P[e] = 0.0;
  }
  if (tid == 0 && a.write_scalars) { double* Ps = P + (size_t)a.K16 * 16 * FT; Ps[0] = 0.0; Ps[1] = 0.0; Ps[2] = 0.0; Ps[3] = 0.0; }
  for (int e = tid; e < B * ZS; e += kWG) zbuf[e] = 0.0;              // the padding columns stay zero: written once

  d4 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  // component k is complete: this wave's tiles -> the partial block.  Register r of lane (q, j) of tile (ti, tj) is the entry
  // (a, b) = (16 ti + 4 r + q, 16 tj + j) of z~' z~: feature (a, b) for a <= b <= Dz
  auto flush = [&](int k, bool first) {
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      if (has[i]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ra = 16 * ti[i] + 4 * r + q, cb = 16 * tj[i] + j;
          if (ra <= cb && cb <= DZ) {
            double* dst = P + (size_t)k * FT + (ra * (DZ + 1) - ra * (ra - 1) / 2 + (cb - ra));
            *dst = first ? acc[i][r] : *dst + acc[i][r];
          }
        }
      }
      acc[i] = d4{0.0, 0.0, 0.0, 0.0};
    }
  };

  bool first = true;
  for (int64_t rg = blockIdx.x; rg < nranges; rg += gridDim.x) {
    const int64_t t0 = rg * R;
    const int nt = (int)(ntiles - t0 < R ? ntiles - t0 : R);
    // ---- the rows of the range by component (thread k = component k): counts, prefix over the components, ids
    wg_sync();
    int cntk = 0;
    if (tid < K) {
#pragma unroll 8
      for (int t = 0; t < nt; ++t) {
        const uint16_t* sg = a.sort_start + (size_t)(t0 + t) * 257 + tid;
        cntk += (int)sg[1] - (int)sg[0];
      }
    }
    int incl = cntk;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
      const int v = __shfl_up(incl, sft);
      if (lane >= sft) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    wg_sync();
    int off = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) off += w < wave ? wsum[w] : 0;
    int o = off + incl - cntk;
    kbase[tid] = o;
    if (tid == kWG - 1) kbase[kWG] = off + incl;
    if (tid < K)
      for (int tb = 0; tb < nt; tb += 4) {
        int s0[4], c[4];
        uint16_t l0[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int t = tb + i < nt ? tb + i : nt - 1;
          const uint16_t* sg = a.sort_start + (size_t)(t0 + t) * 257 + tid;
          s0[i] = sg[0]; c[i] = tb + i < nt ? (int)sg[1] - s0[i] : 0;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int t = tb + i < nt ? tb + i : nt - 1;
          l0[i] = c[i] > 0 ? a.sort_list[(size_t)(t0 + t) * T + s0[i]] : (uint16_t)0;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int t = tb + i;
          if (c[i] > 0) {
            ids[o++] = (uint16_t)((t << 8) | l0[i]);
            const uint16_t* lg = a.sort_list + (size_t)(t0 + t) * T + s0[i];
            for (int m = 1; m < c[i]; ++m) ids[o++] = (uint16_t)((t << 8) | lg[m]);
          }
        }
      }
    wg_sync();
    if (first) {                                         // components without a row in the workgroup's first range: zero rows
      for (int e = tid; e < K * F; e += kWG) {
        const int k = e / F, f = e - k * F;
        if (kbase[k + 1] == kbase[k]) P[(size_t)k * FT + f] = 0.0;
      }
    }
#if MIMO_GRAM_WHATIF == 1
    const int nrows = 0;                                 // (diagnostic: ranges, counts and id lists only)
#else
    const int nrows = kbase[kWG];
#endif
    double gz[2][GPT];                                   // two batches in flight ahead of the one in LDS
    auto fetch = [&](int pos, double (&g)[GPT]) {
      const int nb = nrows - pos < B ? nrows - pos : B;
#pragma unroll
      for (int i = 0; i < GPT; ++i) {
        const int e = tid + kWG * i, r = e / DZ, col = e - r * DZ;
        double v = 0.0;
        if (e < B * DZ && r < nb) {
          const int id = ids[pos + r];
          const int64_t n = (t0 + (id >> 8)) * T + (id & 255);
          v = a.Z[n * DZ + col];
        }
        g[i] = v;
      }
    };
    if (nrows > 0) fetch(0, gz[0]);
    if (nrows > B) fetch(B, gz[1]);
    int k = 0;
    auto step = [&](int pos, double (&g)[GPT]) {
      const int nb = nrows - pos < B ? nrows - pos : B;
#pragma unroll
      for (int i = 0; i < GPT; ++i) {
        const int e = tid + kWG * i, r = e / DZ, col = e - r * DZ;
        if (e < B * DZ) zbuf[r * ZS + col] = g[i];
      }
      for (int e = tid; e < B; e += kWG) zbuf[e * ZS + DZ] = 1.0;
      wg_sync();
      if (pos + 2 * B < nrows) fetch(pos + 2 * B, g);
      int r = 0;
      while (r < nb) {                                   // (uniform control flow: kbase is the same for every thread)
        while (kbase[k + 1] <= pos + r) ++k;
        const int kend = kbase[k + 1] - pos;
        const int rend = kend < nb ? kend : nb;
        // groups of four rows of component k; the lanes whose row lies past the segment read zeros
        // (U groups per iteration, all their operand reads in flight before the first product: one group at a time is an LDS round
        //  trip + a dependent matrix instruction per four rows — ~200 cycles whatever Dz is, measured as a pass whose time did not
        //  depend on Dz: 0.97 - 1.2 ms per 1e7 rows from Dz = 17 to 31)
        constexpr int U = 4;
        for (; r < rend; r += 4 * U) {
          double av[U][TPW], bv[U][TPW];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int rr = r + 4 * u + q;
            const bool inside = rr < rend;
            const double* zr = zbuf + (inside ? rr : 0) * ZS + j;
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
              av[u][i] = inside ? zr[16 * ti[i]] : 0.0;
              bv[u][i] = inside ? zr[16 * tj[i]] : 0.0;
            }
          }
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < TPW; ++i)
              if (has[i]) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][i], bv[u][i], acc[i], 0, 0, 0);
        }
        r = rend;
#if MIMO_GRAM_WHATIF != 2
        if (kend <= nb) flush(k, first);                 // component k is complete
#endif
      }
      wg_sync();
    };
    for (int pos = 0; pos < nrows; pos += 2 * B) {
      step(pos, gz[0]);
      if (pos + B < nrows) step(pos + B, gz[1]);
    }
    first = false;
  }
  if (first) {                                           // a workgroup without a range: an all-zero block
    for (int e = tid; e < K * FT; e += kWG) {
      const int k = e / FT, f = e - k * FT;
      if (f < F) P[(size_t)k * FT + f] = 0.0;
    }
  }
}

static bool label_gram_on() {
  static const bool on = [] { const char* e = getenv("MIMO_LABEL_STATS_GRAM"); return !e || atoi(e) != 0; }();     // tuning knob: 0 = the VALU kernel
  return on;
}
static int g_sorted_range_cap = kSortedRange;
void set_sorted_range_cap(int tiles) { g_sorted_range_cap = tiles < 1 || tiles > kSortedRange ? kSortedRange : tiles; }
static bool label_sorted_on() {
  static const bool on = [] { const char* e = getenv("MIMO_LABEL_STATS_SORTED"); return !e || atoi(e) != 0; }();   // tuning knob
  return on;
}
static int label_sorted_min_d() {
  static const int d = [] { const char* e = getenv("MIMO_LABEL_STATS_SORTED_MIN_D"); return e ? atoi(e) : 17; }();  // tuning knob
  return d;
}
// Dz >= 17; and K > 128 from Dz = 15 (against the two windows of label_stats_wide_kernel, N = 2e6, ms: Dz=16 K=256 0.36 -> 0.27, K=192 0.32 -> 0.25,
// Dz=15 K=160 0.26 -> 0.24; Dz=14 K=200 0.23 -> 0.25 and Dz=13 K=256 0.25 -> 0.26 stay windowed)
static bool label_stats_sorted_covers(int K, int D) {
  if (!label_sorted_on() || D < 10 || D > kMaxD || K < 1 || K > 256) return false;
  return D >= label_sorted_min_d() || (D >= 15 && K > 128);
}
template <int DZ>
static hipError_t launch_sorted(const KernelArgs& a, int grid, hipStream_t stream) {
  hipError_t e = launch_label_tile_sort(a, kLsWideTile, grid, stream);
  if (e != hipSuccess) return e;
  const int64_t ntiles = (a.N + kLsWideTile - 1) / kLsWideTile;
  int R = (int)((ntiles + grid - 1) / (grid > 0 ? grid : 1));            // one range per workgroup where the cap allows (no second round for a few)
  R = R < 1 ? 1 : R > g_sorted_range_cap ? g_sorted_range_cap : R;      // (mimo_tune "sorted_range" lowers the cap: several ranges per workgroup at test sizes)
  const bool gram = label_gram_on();
  if (gram) hipLaunchKernelGGL(label_stats_gram_kernel<DZ>, dim3(grid), dim3(kWG), 0, stream, a, R > gram_range(DZ) ? gram_range(DZ) : R);
  else hipLaunchKernelGGL(label_stats_sorted_kernel<DZ>, dim3(grid), dim3(kWG), 0, stream, a, R);
  return hipGetLastError();
}

bool label_stats_sorted(int K, int D, int structure) { return structure == 0 && label_stats_sorted_covers(K, D); }

static bool xwide_on() {
  static const bool on = [] { const char* e = getenv("MIMO_LABEL_STATS_XWIDE"); return !e || atoi(e) != 0; }();   // tuning knob
  return on;
}
// wide kernel: 4 feature slices up to K = 64, 2 up to K = 128 (Dz <= 12); everything else of the full map: sliced launches
// Dz = 10 .. 16: four feature slices per component up to K = 64, two beyond — in windows of 128 components per launch
// (MIMO_LABEL_STATS_WIDE_ALL=0: the round-2 range — K <= 64, K <= 128 up to Dz = 12 —, the rest on the sliced kernel; tuning knob)
static bool label_stats_wide_covers(int K, int D) {
  static const bool all = [] { const char* e = getenv("MIMO_LABEL_STATS_WIDE_ALL"); return !e || atoi(e) != 0; }();
  return D >= 10 && D <= 16 && (all || K <= 64 || (K <= 128 && D <= 12));
}
static bool label_stats_sorted_covers(int K, int D);
bool label_stats_covers(int K, int D, int structure) {
  if (K < rowwave_min_k() || K > 256 || D < 1) return false;
  if (structure != 0) return D <= 16;            // reduced maps (diagonal / linear): at most 2 Dz + 1 accumulators
  if (D <= 9 || label_stats_wide_covers(K, D)) return true;
  // (K <= 16 at Dz > 16: one launch of 16-thread-per-component slices measured 0.71 against 0.61 ms of the one-hot products,
  //  Dz = 20, K = 16, N = 2e6; from K = 24 on the sliced kernel wins: Dz = 24, K = 24 0.80 / 1.20 ms, Dz = 32, K = 128 1.50 / 2.41 ms)
  if (label_stats_sorted_covers(K, D)) return true;        // (the one-pass kernel takes any K; defined below)
  return xwide_on() && D <= kMaxD && K > 16;
}

// launches of one statistics pass (each reads Z once): 1, or the slice groups of label_stats_xwide_kernel
bool label_stats_sorted(int K, int D, int structure);
int label_stats_launches(int K, int D, int structure) {
  if (label_stats_sorted(K, D, structure)) return 1;
  if (structure == 0 && D > 9 && label_stats_wide_covers(K, D)) return K <= 64 ? 1 : (K + 127) / 128;
  if (structure != 0 || D <= 9) return 1;
  const int fpt = xwide_fpt(D, K);
  int Kp = 1;
  while (Kp < K) Kp <<= 1;
  const int P = kWG / Kp, fpl = P < fpt ? P : fpt;
  return fpt / fpl;
}

// K >= 17 at Dz <= 9 with at least 2^17 rows: label_stats_slots_kernel (MIMO_LABEL_STATS_SLOTS=0: the round-2 kernel, tuning knob)
static bool label_stats_slots_on() {
  static const bool on = [] { const char* e = getenv("MIMO_LABEL_STATS_SLOTS"); return !e || atoi(e) != 0; }();
  return on;
}
bool label_stats_uses_slots(int K, int D, int64_t N) {
  return label_stats_slots_on() && K >= 17 && K <= 256 && D >= 1 && D <= 9 && N >= (1 << 17);
}
size_t label_stats_aux_words() { return kLsAuxWords; }

int label_stats_grid(const KernelArgs& a, int num_cu) {
  const int tile = a.D <= (a.diag ? 10 : 9) ? kLsTile : kLsWideTile;
  const int64_t tiles = (a.N + tile - 1) / tile;
  int64_t g = (int64_t)num_cu * (label_stats_uses_slots(a.K, a.D, a.N) && a.D <= 2 ? 3 : 2);     // (52 KB of LDS, <= 88 registers: three per CU)
  if (!a.diag && a.D >= 10 && label_stats_sorted_covers(a.K, a.D) && label_gram_on()) g = (int64_t)num_cu * gram_wgs_per_cu(a.D);
  if (g > tiles) g = tiles;
  return (int)(g < 1 ? 1 : g);
}

template <int FS>
static void (*pick_label_stats_slots(int D))(const KernelArgs) {
  switch (D) {
    case 1: return label_stats_slots_kernel<1, FS>;   case 2: return label_stats_slots_kernel<2, FS>;
    case 3: return label_stats_slots_kernel<3, FS>;   case 4: return label_stats_slots_kernel<4, FS>;
    case 5: return label_stats_slots_kernel<5, FS>;   case 6: return label_stats_slots_kernel<6, FS>;
    case 7: return label_stats_slots_kernel<7, FS>;   case 8: return label_stats_slots_kernel<8, FS>;
    case 9: return label_stats_slots_kernel<9, FS>;
  }
  return nullptr;
}

template <int FS>
static void (*pick_label_stats_struct(int D))(const KernelArgs) {
  switch (D) {
    case 1: return label_stats_kernel<1, FS>;   case 2: return label_stats_kernel<2, FS>;
    case 3: return label_stats_kernel<3, FS>;   case 4: return label_stats_kernel<4, FS>;
    case 5: return label_stats_kernel<5, FS>;   case 6: return label_stats_kernel<6, FS>;
    case 7: return label_stats_kernel<7, FS>;   case 8: return label_stats_kernel<8, FS>;
    case 9: return label_stats_kernel<9, FS>;
    default: break;
  }
  if constexpr (FS != 0) {          // the reduced maps reach Dz = 16 with one thread per component (the full map: wide kernel)
    switch (D) {
      case 10: return label_stats_kernel<10, FS>; case 11: return label_stats_kernel<11, FS>;
      case 12: return label_stats_kernel<12, FS>; case 13: return label_stats_kernel<13, FS>;
      case 14: return label_stats_kernel<14, FS>; case 15: return label_stats_kernel<15, FS>;
      case 16: return label_stats_kernel<16, FS>;
    }
  }
  return nullptr;
}

// structure: 0 full, 1 diagonal, 2 linear (MIMO_STRUCT_*)
template <int DZ, int FPT>
static hipError_t launch_xwide(const KernelArgs& a, int grid, hipStream_t stream) {
  constexpr int ZS = DZ | 1, T = xwide_tile(FPT);
  const size_t zt = (size_t)(T * ZS > kWG * 8 ? T * ZS : kWG * 8);
  const size_t lds = sizeof(double) * zt + sizeof(uint32_t) * kWG * (T / 32) + sizeof(int) * (2 * kWG + 1 + 4) + sizeof(uint16_t) * T;
  auto fn = label_stats_xwide_kernel<DZ, FPT>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  int Kp = 1;
  while (Kp < a.K) Kp <<= 1;
  const int P = kWG / Kp, FPL = P < FPT ? P : FPT;
  // rank the tiles once for all the launches — from four launches on (profiles/r03_label_presort.txt, N = 2e6, statistics stage in ms without /
  // with: Dz=32 K=128 (8 launches) 2.33 / 1.96, K=256 (16) 4.31 / 3.64, Dz=20 K=96 (4) 0.76 / 0.71, Dz=24 K=128 (4) 0.89 / 0.85; two launches
  // lose the extra pass: Dz=32 K=64 0.95 / 1.01): the ranking is a small part of a launch, the z tile and the member loop are the rest
  const bool presort = FPT / FPL >= 4 && label_presort_on() && a.sort_list && a.sort_start;
  if (presort) {
    e = launch_label_tile_sort(a, T, grid, stream);
    if (e != hipSuccess) return e;
  }
  for (int s0 = 0; s0 < FPT; s0 += FPL) {          // FPT / FPL launches into the same partial block
    KernelArgs g = a;
    g.presort = presort ? 1 : 0;
    g.cb0 = s0;
    if (s0 > 0) g.write_scalars = 0;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kWG), lds, stream, g);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// zero the histogram in front of a label kernel that counts its own labels (KernelArgs::fuse_hist)
hipError_t launch_label_hist_reset(const KernelArgs& a, hipStream_t stream) {
  return a.aux ? hipMemsetAsync(a.aux, 0, 256 * sizeof(uint32_t), stream) : hipErrorInvalidValue;
}
// the label pass of (K, F16) runs on gibbs_rowwave_kernel (Theta resident), the kernel that can count its labels
bool gibbs_rowwave_counts_labels(int K, int F16, int ZS) { return rowwave_resident(K, F16, ZS); }

// structure: 0 full, 1 diagonal, 2 linear (MIMO_STRUCT_*)
hipError_t launch_label_stats(const KernelArgs& a, int structure, int grid, hipStream_t stream) {
  typedef void (*fn_t)(const KernelArgs);
  if (a.D < 1 || a.D > kMaxD || a.K < 1 || a.K > 256) return hipErrorInvalidValue;
  fn_t fn = nullptr;
  if (label_stats_uses_slots(a.K, a.D, a.N)) {
    if (!a.aux) return hipErrorInvalidValue;
    fn = structure == 1 ? pick_label_stats_slots<1>(a.D) : structure == 2 ? pick_label_stats_slots<2>(a.D) : pick_label_stats_slots<0>(a.D);
    if (!fn) return hipErrorInvalidValue;
    if (!a.fuse_hist) {       // (else: the label kernel of this pass counted them, launch_label_hist_reset ran in front of it)
      hipError_t e = hipMemsetAsync(a.aux, 0, 256 * sizeof(uint32_t), stream);
      if (e != hipSuccess) return e;
      int hg = (int)((a.N + kWG * 16 - 1) / (kWG * 16));
      if (hg > 1024) hg = 1024;
      hipLaunchKernelGGL(label_hist_kernel, dim3(hg), dim3(kWG), 0, stream, a.labels, a.N, a.K, a.aux);
    }
    hipLaunchKernelGGL(label_slots_kernel, dim3(1), dim3(kWG), 0, stream, a.aux, a.K);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kWG), 0, stream, a);
    return hipGetLastError();
  }
  if (structure == 0 && a.sort_list && a.sort_start && label_stats_sorted_covers(a.K, a.D)) {
    switch (a.D) {
#define MIMO_SD(d) case d: return launch_sorted<d>(a, grid, stream);
      MIMO_SD(10) MIMO_SD(11) MIMO_SD(12) MIMO_SD(13) MIMO_SD(14) MIMO_SD(15) MIMO_SD(16) MIMO_SD(17) MIMO_SD(18) MIMO_SD(19) MIMO_SD(20)
      MIMO_SD(21) MIMO_SD(22) MIMO_SD(23) MIMO_SD(24) MIMO_SD(25) MIMO_SD(26) MIMO_SD(27) MIMO_SD(28) MIMO_SD(29) MIMO_SD(30) MIMO_SD(31) MIMO_SD(32)
#undef MIMO_SD
    }
  }
  if (structure == 1) fn = pick_label_stats_struct<1>(a.D);
  else if (structure == 2) fn = pick_label_stats_struct<2>(a.D);
  else if (a.D <= 9) fn = pick_label_stats_struct<0>(a.D);
  else if (label_stats_wide_covers(a.K, a.D)) {
    static const fn_t wide4[7] = {label_stats_wide_kernel<10, 4>, label_stats_wide_kernel<11, 4>, label_stats_wide_kernel<12, 4>,
                                  label_stats_wide_kernel<13, 4>, label_stats_wide_kernel<14, 4>, label_stats_wide_kernel<15, 4>,
                                  label_stats_wide_kernel<16, 4>};
    static const fn_t wide2[7] = {label_stats_wide_kernel<10, 2>, label_stats_wide_kernel<11, 2>, label_stats_wide_kernel<12, 2>,
                                  label_stats_wide_kernel<13, 2>, label_stats_wide_kernel<14, 2>, label_stats_wide_kernel<15, 2>,
                                  label_stats_wide_kernel<16, 2>};
    fn = a.K <= 64 ? wide4[a.D - 10] : wide2[a.D - 10];
    if (a.K > 128) {                  // windows of 128 components, one launch each into the same partial block
      const bool presort = label_presort_on() && a.sort_list && a.sort_start;
      if (presort) {
        hipError_t e = launch_label_tile_sort(a, kLsWideTile, grid, stream);
        if (e != hipSuccess) return e;
      }
      for (int k0 = 0; k0 < a.K; k0 += 128) {
        KernelArgs w = a;
        w.presort = presort ? 1 : 0;
        w.k0 = k0;
        hipLaunchKernelGGL(fn, dim3(grid), dim3(kWG), 0, stream, w);
      }
      return hipGetLastError();
    }
  } else {
    const bool s8 = xwide_fpt(a.D, a.K) == 8;
    switch (a.D) {
#define MIMO_XW4(d) case d: return launch_xwide<d, 4>(a, grid, stream);
#define MIMO_XW8(d) case d: return launch_xwide<d, 8>(a, grid, stream);
#define MIMO_XWB(d) case d: return s8 ? launch_xwide<d, 8>(a, grid, stream) : launch_xwide<d, 16>(a, grid, stream);
      MIMO_XW4(10) MIMO_XW4(11) MIMO_XW4(12) MIMO_XW4(13) MIMO_XW4(14) MIMO_XW4(15) MIMO_XW4(16)
      MIMO_XW8(17) MIMO_XW8(18) MIMO_XW8(19) MIMO_XW8(20) MIMO_XW8(21) MIMO_XW8(22) MIMO_XW8(23) MIMO_XW8(24) MIMO_XW8(25)
      MIMO_XW8(26) MIMO_XW8(27) MIMO_XW8(28) MIMO_XW8(29)
      MIMO_XWB(30) MIMO_XWB(31) MIMO_XWB(32)
#undef MIMO_XW4
#undef MIMO_XW8
#undef MIMO_XWB
      default: return hipErrorInvalidValue;
    }
  }
  if (!fn) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(kWG), 0, stream, a);
  return hipGetLastError();
}

}  // namespace mimo
