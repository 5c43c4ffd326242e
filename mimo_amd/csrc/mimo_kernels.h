// Internal interface between the C-ABI layer (mimo_abi.cpp) and the gfx950 kernels
// (mimo_kernels.hip).  Not installed; the public surface is include/mimo_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mimo {

constexpr int kWG = 256;        // threads per workgroup (4 wavefronts of 64)
constexpr int kTile = 32;       // data rows per tile
constexpr int kMaxNCB = 10;     // 16-wide feature column blocks one launch accumulates
constexpr int kMaxFusedD = 16;  // largest Dz the single-pass fused kernels cover (F = 153 features)
constexpr int kMaxD = 32;       // largest Dz overall (two-stage path: chunked E-step + statistics per column group)
constexpr int kChunkNCB = 9;    // feature column blocks per chunk of the chunked E-step (144 features; Dz=32 -> 4 chunks)

// Where the per-tile weight table R (rows x K) comes from.
enum Source : int { kSrcEstep = 0, kSrcWeights = 1, kSrcLabels = 2 };
// Compile-time specialisations of the fused kernel: the two hot modes carry no optional-output or
// other-mode code (register pressure decides occupancy here); kGeneric keeps every runtime flag.
enum Mode : int { kFastVI = 0, kFastGibbs = 1, kGeneric = 2, kModeWeights = 3, kModeLabels = 4 };

constexpr int kThetaInline = 40;

struct KernelArgs {
  const double* Z;        // (N, D) row-major observations
  int64_t N;
  int D;                  // data dimension Dz
  int K;                  // true number of components
  int K16;                // ceil(K / 16) row blocks
  int F16;                // padded feature count (multiple of 16)
  int ZS, RS, LS;         // LDS row strides (doubles) of the z / feature / weight tiles
  const double* theta;    // MFMA A-operand image of the parameter block: [K16][F16/4][64]
  const uint8_t* feat;    // [F16][2] index pairs (a,b) into z~ = [z, 1, 0]
  double* partials;       // [G][K16*16*F16 + 4] per-workgroup partial statistics + scalars
  double* resp;           // (K,N) table: output of the E-step / input of kSrcWeights; may be null
  double* logp;           // (K,N) l[k,n] output; may be null
  double* lse;            // (N,) output; may be null
  int32_t* labels;        // (N,) output of the Gibbs draw / input of kSrcLabels; may be null
  const double* u;        // (N,) uniforms or null (=> in-kernel Philox)
  uint64_t seed, sweep;
  int64_t row0;
  int gibbs;              // 0: softmax responsibilities, 1: categorical draw (one-hot weights)
  int do_stats;
  int split;              // also accumulate sum_k r l (entropy split of the ELBO scalars)
  int64_t ntiles;
  int cb0;                // statistics modes: first 16-wide feature column block of this launch (label_stats_xwide_kernel: first feature slice)
  int F16_total;          // padded feature count of the whole problem (partials row stride)
  int write_scalars;      // write the 4 scalar slots of the partial block (0: another launch owns them)
  int diag;               // feature table is a reduced one (diagonal: 2 Dz + 1, linear: Dz + 1 features): table-driven E-step kernels
  unsigned long long* stamps;  // diagnostic builds (-DMIMO_STAMPS) only: [grid][4 waves][8] phase cycle sums
  double theta_inline[40];  // small-shape kernel with one lane per row (G = 1): Theta itself (kThetaInline doubles at most)
  uint32_t* aux;          // label_stats_slots_kernel: label histogram + slot table (label_stats_aux_words() words)
  uint16_t* sort_list;    // label_tile_sort_kernel -> label_stats_wide / _xwide_kernel: per tile the rows in component order, [ntiles][T]
  uint16_t* sort_start;   // ... and the first list position of every component, [ntiles][257] (entry 256: rows in the list)
  int presort;            // the statistics launch reads sort_list / sort_start instead of ranking the tile's labels itself
  int k0;                 // label_stats_wide_kernel: first component of the launch's window (components k0 .. k0 + 127 of K > 128)
  int fuse_hist;          // gibbs_rowwave_kernel counts the labels it draws into aux[0 .. 255] (no label_hist_kernel behind it)
};

// feature count helpers (z~ = [z,1]; features = upper-triangular pairs of z~)
inline int feat_count(int D) { return (D + 1) * (D + 2) / 2; }
inline int feat_pad16(int D) { return (feat_count(D) + 15) / 16 * 16; }
inline int feat_index(int D, int a, int b) {  // a <= b <= D
  return a * (D + 1) - a * (a - 1) / 2 + (b - a);
}
// diagonal structure (W_k diagonal): only the 2D+1 features z_a^2, z_a, 1 exist —
//   f = a for (a,a), D + a for (a,D), 2D for (D,D)
inline int diag_feat_count(int D) { return 2 * D + 1; }
inline int diag_feat_pad16(int D) { return (diag_feat_count(D) + 15) / 16 * 16; }
inline int diag_feat_index(int D, int a, int b) { return a == D ? 2 * D : (b == D ? D + a : a); }
// linear structure (all W_k equal: the quadratic term is common to every component and leaves the softmax):
// only the D+1 features z_a, 1 exist — f = a for (a,D), D for (D,D)
inline int lin_feat_count(int D) { return D + 1; }
inline int lin_feat_pad16(int D) { return (lin_feat_count(D) + 15) / 16 * 16; }

size_t fused_lds_bytes(const KernelArgs& a, int src);
int fused_grid(const KernelArgs& a, int num_cu, int src);
// returns hipSuccess or an error; sets *unsupported when (K, D, src) has no kernel
hipError_t launch_fused(const KernelArgs& a, int src, int grid, hipStream_t stream, bool* unsupported);
// chunked E-step for shapes outside the fused kernels (Dz > 16, or K > 64 with Dz > 9): no statistics
size_t chunked_lds_bytes(const KernelArgs& a);
// contraction steps per row block of the Theta image of the two-stage path: padded to whole chunks of BOTH E-step kernels
// (estep_chunked_kernel: 36 steps per chunk; wide_estep_kernel: 24)
__host__ __device__ inline int chunked_ns_pad(int F16) { return (F16 + 287) / 288 * 72; }
hipError_t launch_estep_chunked(const KernelArgs& a, int grid, hipStream_t stream);
bool fused_covers(int K16, int ncb, int src);
int stats_group_ncb(int K16);   // feature column blocks one statistics launch can accumulate for this K
hipError_t launch_reduce(const double* partials, int G, int64_t stride, double* out, hipStream_t stream);
// mask_structure: 0 = write every feature of the table; MIMO_STRUCT_DIAG / _LINEAR = the table is the FULL map but
// only the entries of that structure are real (small-shape kernel): the others come back as zeros
// both in one launch (mimo_small.hip)
hipError_t launch_reduce_unpack(const double* partials, int G, int64_t stride, const uint8_t* feat, int K, int D, int F, int F16,
                                double* S_packed, double* scalars3, hipStream_t stream, int mask_structure = 0);
hipError_t launch_unpack(const double* reduced, const uint8_t* feat, int K, int D, int F, int F16,
                         double* S_packed, double* scalars3, hipStream_t stream, int mask_structure = 0);

// Small shapes (Dz <= 4, K <= 32): float64 VALU kernel, one datum per lane (mimo_small.hip).  theta: [G KL][F]
// row-major over the FULL feature map in feat_index order; partial blocks in the layout of the tile kernels with
// F16_total = 16.
constexpr int kSmallMaxD = 4;
constexpr int kSmallMaxK = 32;
int small_kl(int D, int K);     // components per lane
int small_g(int D, int K);      // lanes per row
bool small_covers(int D, int K);
int small_grid(const KernelArgs& a, int num_cu, int src);
hipError_t launch_small(const KernelArgs& a, int src, int grid, hipStream_t stream, bool* unsupported);

// Large-K Gibbs sweep (mimo_rowwave.hip): row-owner label kernel (Theta stationary in LDS, draw in registers) and
// the label-indexed statistics kernel.  theta: [NS][KB][64] with the component permutation of rowwave_component().
size_t rowwave_lds_bytes(int KB, int NS, int ZS);
int rowwave_kb(int K);
int rowwave_kb_shape(int K, int F16, int ZS);     // ... for the shape (the streamed walk may take two more than rowwave_kb)
bool rowwave_covers(int K, int F16, int ZS);
int rowwave_image_ns(int K, int F16, int ZS);     // contraction steps of the operand image (padded to whole chunks where Theta streams)
int rowwave_grid(const KernelArgs& a, int num_cu);
hipError_t launch_gibbs_rowwave(const KernelArgs& a, int grid, hipStream_t stream);
bool label_stats_covers(int K, int D, int structure);
int label_stats_grid(const KernelArgs& a, int num_cu);
hipError_t launch_label_stats(const KernelArgs& a, int structure, int grid, hipStream_t stream);

hipError_t launch_table_entropy(const double* table, int64_t count, double* partials, int nblocks,
                                double* out, hipStream_t stream);

// host mirror of the in-kernel counter-based generator
double philox_uniform_host(uint64_t seed, uint64_t row, uint64_t sweep);

// Posterior-predictive mixture moments (mimo_predict): one thread per row of Z.
struct PredictArgs {
  const double* Z; int64_t N; int dx, dc, dy, K, mode;   // dc = dx + affine
  const double* gate;   // [K][1 + dx + dx*dx]  canonical (c, b, W) of the log-weights
  const double* M;      // [K][dy][dc]
  const double* Q;      // [K][dc][dc]   cs = 1 + x~' Q x~
  const double* Cc;     // [K][dy][dy]   covar = cs * Cc
  const double* y;      // [N][dy] or null
  const double* P;      // [K][dy][dy]   precision at cs = 1 (for the predictive log-density), or null
  const double* ld;     // [K]           logdet P
  double* mu; double* covar; double* nlpd;
  int diag;             // 1: covar receives the variances [N][dy] followed by the standard deviations [N][dy] (MIMO_F_DIAG_VAR)
};
constexpr int kMaxPredictDy = 8;
constexpr double kPadLogDensity = -1e300;   // c_k of the padding components k in [K, 16*K16)
constexpr double kOffLogDensity = -1e299;   // l below this: a padding or switched-off component (its l is -1e300 to the last bit)
hipError_t launch_predict(const PredictArgs& a, hipStream_t stream, bool* unsupported);
bool launch_predict_reg(const PredictArgs& a, hipStream_t stream, hipError_t* err);      // mimo_predict.hip: dx <= 8 with the affine column

}  // namespace mimo
