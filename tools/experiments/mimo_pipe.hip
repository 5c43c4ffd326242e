// EXPERIMENT, not part of libmimo_hip.so (negative result, kept for the record; DESIGN.md section 4, "what did not work").
// Measured on MI355X, N = 1e7: C2 shape (Dz = 16, K = 64) 7.15 - 7.40 ms against 6.81 - 6.86 ms of fused_kernel; Dz = 12, K = 64
// 4.75 - 4.93 against 4.36 - 4.41 ms.  The same structure WITHOUT any feature build runs in 6.44 ms: the hooks cost 0.7 - 0.9 ms
// whatever their VALU count (5 instructions per feature with selected addresses, or 1 with half-wave variants and immediate
// offsets) — the matrix phases of this kernel already keep the LDS pipe half busy with operand reads (one 512-byte read
// per MFMA and wave), and the build's reads and writes queue in front of them (LDS returns in order).  In fused_kernel the
// build has the LDS pipe to itself.  To try it again: copy next to mimo_kernels.hip, add to the Makefile, and let
// resolve_fused() ask pick_fused_pipe() first for mode == kFastVI.
//
// The plain mean-field pass (softmax + statistics, no tables) of K <= 64, Dz = 5 .. 16 — fused_kernel<NCB, 1, kFastVI, DS>
// with its feature build moved under the matrix phases.
//
// What the phase trace of fused_kernel shows at C2 (tools/stamps.py): the two workgroups of a CU run their tiles in step,
// and the feature build (z~ rows -> registers -> 20 products -> LDS, 10 - 18 % of a wave's tile time) is a latency-bound
// phase both sit in together with the matrix pipe idle.  A second feature tile would let the next tile's build hide under
// this tile's MFMAs, but 2 x 41 KB do not fit LDS next to two workgroups per CU.  The feature tile is therefore refilled
// in two halves, each as soon as its columns are dead:
//
//   P1  L  = Theta . Phi'   contraction steps over columns [0, H)      | hook: build columns [H, F16) of THIS tile
//   -- barrier --                                                      |       (dead since the previous tile's P5)
//   P2  L += ...            contraction steps over columns [H, F16); L -> LDS
//   -- barrier --  P3  softmax over k (normalise_tile)  -- barrier --
//   P4  S += R . Phi        column blocks of [0, H)                    | hook: z~ rows of the NEXT tile -> LDS
//   -- barrier --                                                      |
//   P5  S += R . Phi        column blocks of [H, F16)                  | hook: build columns [0, H) of the NEXT tile
//   -- barrier --
//
// Five barriers per tile instead of four, no extra LDS, the same products in the same order (bit-identical statistics),
// and no phase left in which a wave waits out LDS round trips with nothing else to issue: the z~ row is read into
// registers after one matrix step and multiplied after a later one (in groups of five features per lane).
#include "mimo_tile.h"
#include "mimo_extra.h"

#include <cstdlib>

namespace mimo {

// lane (row = lane & 31, parity = lane >> 5) of wave W makes features BASE + W*FW + 2i + parity, i < FW/2, in groups of
// at most 5: the operands of a group are read from the z~ row in LDS — the two half-waves in turn, each with compile-time
// offsets (immediates of the ds_read: no address arithmetic on the pipe the MFMAs use) — then multiplied and stored.
// read_group and store_group sit behind different matrix steps.
template <int D, int BASE, int FW, int W, int I0, int P, int... I>
__device__ __forceinline__ void read_group(const double* __restrict__ zrow, double (&za)[5], double (&zb)[5],
                                           std::integer_sequence<int, I...>) {
  ((za[I] = zrow[FeatAB<D, BASE + W * FW + 2 * (I0 + I) + P>::a], zb[I] = zrow[FeatAB<D, BASE + W * FW + 2 * (I0 + I) + P>::b]), ...);
}
template <int BASE, int FW, int W, int I0, int... I>
__device__ __forceinline__ void store_group(double* __restrict__ prow, const double (&za)[5], const double (&zb)[5],
                                            std::integer_sequence<int, I...>) {
  ((prow[BASE + W * FW + 2 * (I0 + I)] = za[I] * zb[I]), ...);
}

template <int NCB, int DS>
__global__ __launch_bounds__(kWG, 2) void fused_pipe_kernel(const KernelArgs a) {
  constexpr int T = kTile, NSI = 4 * NCB;
  constexpr int NS = ((DS + 1) * (DS + 2) / 2 + 3) / 4;      // contraction steps that carry features (the padded tail is skipped)
  constexpr int NCB0 = (NCB + 1) / 2, NCB1 = NCB - NCB0;      // column blocks of the two halves
  constexpr int NS0 = 4 * NCB0 < NS ? 4 * NCB0 : NS;          // steps over the first half
  static_assert(NCB >= 2 && NS > NS0, "two non-empty halves");

  extern __shared__ __align__(16) unsigned char smem[];
  double* Zs = reinterpret_cast<double*>(smem);  // [T][ZS]   z~ rows (z, 1, 0)
  double* Ph = Zs + T * a.ZS;                    // [T][RS]   feature tile
  double* Lt = Ph + T * a.RS;                    // [T][LS]   l -> e -> r per (row, component)
  double* red = Lt + T * a.LS;                   // [16]      block-reduction scratch
  double* etab = red + 16;                       // [64]      2^(j/64) for exp_nonpos
  int* labs = reinterpret_cast<int*>(red);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int D = a.D, K = a.K, K16 = a.K16;
  const int ZS = a.ZS, RS = a.RS, LS = a.LS;
  const int Kpad = K16 * 16;
  const int64_t N = a.N, G = gridDim.x;
  if (tid < 64) etab[tid] = exp2((double)tid * (1.0 / 64.0));

  // Theta stream (as fused_kernel): the NS slices of this wave's row block through an 8-deep register ring that wraps
  // into the next tile; padded to a multiple of the ring depth so that element e always sits in slot e % PF
  constexpr int PF = NS < 8 ? NS : 8, NEP = (NS + PF - 1) / PF * PF;
  const gptr_t thw0 = (gptr_t)(a.theta + (size_t)(wave < K16 ? wave : 0) * NSI * 64);
  gptr_t thw = thw0;
  auto theta_slice = [&](int e) -> double { return thw[(e >= NS ? 0 : e) * 64 + lane]; };
  double ring[PF];
#pragma unroll
  for (int e = 0; e < PF; ++e) ring[e] = theta_slice(e);

  d4 sacc[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) sacc[cb] = d4{0.0, 0.0, 0.0, 0.0};
  double sc_lse = 0.0, sc_rl = 0.0, sc_prod = 1.0;
  int prod_tiles = 0;
  PhiloxBatch pbatch;

  // z staging: T*D <= 512 elements, 2 per thread; the tile after next travels in registers
  int zoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + kWG * i, pt = e / D;
    zoff[i] = e < T * D ? pt * ZS + (e - pt * D) : -1;
  }
  double zr[2];
  auto load_z = [&](int64_t t) {
    const int64_t base = t * T * D, total = N * D;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t g = base + tid + kWG * i;
      zr[i] = (zoff[i] >= 0 && g < total) ? a.Z[g] : 0.0;
    }
  };
  auto store_z = [&](int64_t t) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (zoff[i] >= 0) Zs[zoff[i]] = zr[i];
    if (tid < T) {
      Zs[tid * ZS + D] = (t * T + tid) < N ? 1.0 : 0.0;  // rows past N contribute nothing
      Zs[tid * ZS + D + 1] = 0.0;                        // padded features read this slot
    }
  };

  // feature build of one half: the z~ row of this lane's datum goes to registers (read_row), the products of the half
  // are written later (build_half): lane (row = lane & 31, parity = lane >> 5) of wave W makes features
  // BASE + W*FW + 2i + parity
  const int frow = tid & (T - 1);
  const bool parity = (lane >> 5) != 0;
  double za[5], zb[5];
  // group g (5 features per lane) of half H: operand reads / products + stores
  auto group_reads = [&](auto half_c, auto g_c) {
    constexpr int H = decltype(half_c)::value, GI = decltype(g_c)::value;
    constexpr int BASE = H == 0 ? 0 : 16 * NCB0, FW = 4 * (H == 0 ? NCB0 : NCB1), NI = FW / 2;
    constexpr int I0 = 5 * GI, CNT = NI - I0 < 5 ? NI - I0 : 5;
    if constexpr (CNT > 0) {
      int zoff_row = frow * ZS;
      asm volatile("" : "+v"(zoff_row));      // opaque per call: no loop-invariant address registers across the tile loop
      const double* zrow = Zs + zoff_row;
      using Seq = std::make_integer_sequence<int, CNT>;
      if (parity) {
        switch (wave) {   // scalar: no divergence
          case 0: read_group<DS, BASE, FW, 0, I0, 1>(zrow, za, zb, Seq{}); break;
          case 1: read_group<DS, BASE, FW, 1, I0, 1>(zrow, za, zb, Seq{}); break;
          case 2: read_group<DS, BASE, FW, 2, I0, 1>(zrow, za, zb, Seq{}); break;
          default: read_group<DS, BASE, FW, 3, I0, 1>(zrow, za, zb, Seq{}); break;
        }
      } else {
        switch (wave) {
          case 0: read_group<DS, BASE, FW, 0, I0, 0>(zrow, za, zb, Seq{}); break;
          case 1: read_group<DS, BASE, FW, 1, I0, 0>(zrow, za, zb, Seq{}); break;
          case 2: read_group<DS, BASE, FW, 2, I0, 0>(zrow, za, zb, Seq{}); break;
          default: read_group<DS, BASE, FW, 3, I0, 0>(zrow, za, zb, Seq{}); break;
        }
      }
    }
  };
  auto group_stores = [&](auto half_c, auto g_c) {
    constexpr int H = decltype(half_c)::value, GI = decltype(g_c)::value;
    constexpr int BASE = H == 0 ? 0 : 16 * NCB0, FW = 4 * (H == 0 ? NCB0 : NCB1), NI = FW / 2;
    constexpr int I0 = 5 * GI, CNT = NI - I0 < 5 ? NI - I0 : 5;
    if constexpr (CNT > 0) {
      int poff_row = frow * RS + (lane >> 5);
      asm volatile("" : "+v"(poff_row));
      double* prow = Ph + poff_row;
      using Seq = std::make_integer_sequence<int, CNT>;
      switch (wave) {
        case 0: store_group<BASE, FW, 0, I0>(prow, za, zb, Seq{}); break;
        case 1: store_group<BASE, FW, 1, I0>(prow, za, zb, Seq{}); break;
        case 2: store_group<BASE, FW, 2, I0>(prow, za, zb, Seq{}); break;
        default: store_group<BASE, FW, 3, I0>(prow, za, zb, Seq{}); break;
      }
    }
  };
  constexpr int NG0 = (2 * NCB0 + 4) / 5, NG1 = (2 * NCB1 + 4) / 5;     // groups per half (2 NCBh features per lane)
  static_assert(NG0 <= 2 && NG1 <= 2, "at most two groups per half: four hook slots");

  // prologue: z~ of the first tile, first half of its features
  load_z(blockIdx.x);
  store_z(blockIdx.x);
  load_z(blockIdx.x + G);
  wg_sync();
  group_reads(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  group_stores(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  group_reads(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
  group_stores(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
  wg_sync();

  for (int64_t t = blockIdx.x; t < a.ntiles; t += G) {
    const int64_t n0 = t * T;
    thw = thw0;
    asm volatile("" : "+s"(thw));  // opaque per tile: slice addresses = scalar base + immediates, not NS hoisted VGPR pairs
    d4 acc[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
    const double* p0 = Ph + j * RS + q;
    const double* p1 = Ph + (16 + j) * RS + q;
    // contraction steps [S0, S1) of L = Theta . Phi' (B operands two steps ahead of their MFMAs), with the feature
    // build of the second half hooked behind steps HR (row read) and HB (products)
    auto estep_part = [&](auto s0c, auto s1c, auto hooks_c) {
      constexpr int S0 = decltype(s0c)::value, S1 = decltype(s1c)::value;
      constexpr bool HOOKS = decltype(hooks_c)::value;
      // hook slots: reads of group 0 after step H0, its stores two steps later, then group 1 likewise
      constexpr int H0 = S0 + 1, H1 = S0 + 3, H2 = S0 + 5, H3 = S0 + 7;
      static_assert(!HOOKS || H3 < S1, "four hook slots inside the part");
      auto hook = [&](int s) {
#ifdef MIMO_PIPE_WHATIF_NOBUILD
        return;
#endif
        if (HOOKS && s == H0) { group_reads(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}); __builtin_amdgcn_sched_barrier(0); }
        if (HOOKS && s == H1) { group_stores(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}); __builtin_amdgcn_sched_barrier(0); }
        if (HOOKS && s == H2) { group_reads(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}); __builtin_amdgcn_sched_barrier(0); }
        if (HOOKS && s == H3) { group_stores(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}); __builtin_amdgcn_sched_barrier(0); }
      };
      if (wave < K16) {      // ONE wave-uniform branch around the whole part: straight-line steps, loads stay in flight
        double bq0[3], bq1[3];
        bq0[S0 % 3] = p0[4 * S0]; bq1[S0 % 3] = p1[4 * S0];
        if (S0 + 1 < S1) { bq0[(S0 + 1) % 3] = p0[4 * (S0 + 1)]; bq1[(S0 + 1) % 3] = p1[4 * (S0 + 1)]; }
#pragma unroll
        for (int s = S0; s < S1; ++s) {
          if (s + 2 < S1) { bq0[(s + 2) % 3] = p0[4 * (s + 2)]; bq1[(s + 2) % 3] = p1[4 * (s + 2)]; }
          __builtin_amdgcn_sched_barrier(0);
          const double av = ring[s % PF];
          ring[s % PF] = theta_slice((s + PF) % NEP);      // wraps into the next tile's first slices
          acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bq0[s % 3], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bq1[s % 3], acc[1], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          hook(s);
        }
      } else {               // a wave without a row block (K <= 48) only takes its share of the feature build
#pragma unroll
        for (int s = S0; s < S1; ++s) hook(s);
      }
    };
    // ---- P1: first half of the contraction; second half of this tile's features is built underneath
    estep_part(std::integral_constant<int, 0>{}, std::integral_constant<int, NS0>{}, std::true_type{});
    wg_sync();
    // ---- P2: second half; L tile -> LDS (C layout: reg r of lane (q, j) = component q + 4r, datum j)
    estep_part(std::integral_constant<int, NS0>{}, std::integral_constant<int, NS>{}, std::false_type{});
    if (wave < K16) {
#pragma unroll
      for (int e = NS; e < NEP; ++e) ring[e % PF] = theta_slice((e + PF) % NEP);   // stream padding
      int lw_off = j * LS + 16 * wave + q;
      asm volatile("" : "+v"(lw_off));
      double* lw0 = Lt + lw_off;
      double* lw1 = lw0 + 16 * LS;
#pragma unroll
      for (int r = 0; r < 4; ++r) { lw0[4 * r] = acc[0][r]; lw1[4 * r] = acc[1][r]; }
    }
    wg_sync();
    // ---- P3: softmax over k (8 lanes per datum)
    __builtin_amdgcn_s_setprio(2);
    normalise_tile<1, kFastVI>(a, Lt, LS, etab, K, K16, N, n0, wave, lane, false, nullptr, nullptr, nullptr,
                               sc_lse, sc_rl, sc_prod, labs, pbatch, G * T);
    if (++prod_tiles == 64) {   // K^64 <= 256^64 = 2^512 stays inside the float64 range
      sc_lse += log(sc_prod);
      sc_prod = 1.0;
      prod_tiles = 0;
    }
    __builtin_amdgcn_s_setprio(0);
    wg_sync();
    // ---- P4 / P5: S += R . Phi over the column blocks of one half: step s contracts rows {s, s+8, s+16, s+24};
    //      A lane (i = j, kk = q) = R[8q+s][16 wave + j], B lane (kk = q, col j) = Phi[8q+s][16cb + j]
    int lt_off = 8 * q * LS + 16 * wave + j, ph_off = 8 * q * RS + j;
    asm volatile("" : "+v"(lt_off), "+v"(ph_off));
    const double* ltq = Lt + lt_off;
    const double* phq = Ph + ph_off;
    auto stats_half = [&](auto cb0c, auto ncbc, auto hook) {
      constexpr int CB0 = decltype(cb0c)::value, NC = decltype(ncbc)::value;
      if (wave < K16) {
        double avq[2], bvq[2][NC];
        auto fetch = [&](int s, int slot) {
          avq[slot] = ltq[s * LS];
#pragma unroll
          for (int cb = 0; cb < NC; ++cb) bvq[slot][cb] = phq[s * RS + 16 * (CB0 + cb)];
        };
        fetch(0, 0);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          if (s + 1 < 8) fetch(s + 1, (s + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int cb = 0; cb < NC; ++cb)
            sacc[CB0 + cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(avq[s & 1], bvq[s & 1][cb], sacc[CB0 + cb], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          hook(s);
        }
      } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) hook(s);
      }
    };
    // P4: first half; the next tile's z~ rows go to LDS underneath (Zs was last read in P1)
    stats_half(std::integral_constant<int, 0>{}, std::integral_constant<int, NCB0>{}, [&](int s) {
      if (s == 1) {
        store_z(t + G);
        load_z(t + 2 * G);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    wg_sync();
    // P5: second half; the first half of the NEXT tile's features is built underneath (its columns are dead)
    stats_half(std::integral_constant<int, NCB0>{}, std::integral_constant<int, NCB1>{}, [&](int s) {
#ifdef MIMO_PIPE_WHATIF_NOBUILD
      return;
#endif
      if (s == 1) { group_reads(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}); __builtin_amdgcn_sched_barrier(0); }
      if (s == 3) { group_stores(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}); __builtin_amdgcn_sched_barrier(0); }
      if (s == 4) { group_reads(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}); __builtin_amdgcn_sched_barrier(0); }
      if (s == 6) { group_stores(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}); __builtin_amdgcn_sched_barrier(0); }
    });
    wg_sync();
  }

  // ---- per-workgroup partials (layout of fused_kernel)
  const int FT = a.F16_total;
  const size_t pstride = (size_t)Kpad * FT + 4;
  double* P = a.partials + (size_t)blockIdx.x * pstride;
  if (wave < K16) {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) P[(size_t)(16 * wave + q + 4 * r) * FT + 16 * cb + j] = sacc[cb][r];
  }
  sc_lse += log(sc_prod);
  sc_lse = wave_sum(sc_lse);
  sc_rl = wave_sum(sc_rl);
  wg_sync();
  if (lane == 0) { red[2 * wave] = sc_lse; red[2 * wave + 1] = sc_rl; }
  wg_sync();
  if (tid == 0 && a.write_scalars) {
    double* Ps = a.partials + (size_t)blockIdx.x * pstride + (size_t)Kpad * FT;
    Ps[0] = (red[0] + red[2]) + (red[4] + red[6]);
    Ps[1] = (red[1] + red[3]) + (red[5] + red[7]);
    Ps[2] = 0.0;     // no entropy split in this mode
    Ps[3] = 0.0;
  }
}

// ------------------------------------------------------------------------------------------
// selection (resolve_fused of mimo_kernels.hip asks here first for the plain mean-field pass)
// ------------------------------------------------------------------------------------------
typedef void (*pipe_fn)(const KernelArgs);
pipe_fn pick_fused_pipe(int D, int K16) {
  static const bool on = [] { const char* e = getenv("MIMO_PIPE"); return !e || atoi(e) != 0; }();   // tuning knob
  if (!on || K16 < 3 || K16 > 4) return nullptr;
  switch (D) {
    case 12: return fused_pipe_kernel<6, 12>;
    case 16: return fused_pipe_kernel<10, 16>;
    default: return nullptr;
  }
}

}  // namespace mimo
