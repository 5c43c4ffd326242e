// Launchers of the helper kernels that are not part of the tile-kernel interface (kept out of mimo_kernels.h so that
// adding one does not rebuild every instantiation of the tile kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mimo {

// categorical draw from a (K, N) table of log-probabilities (mimo_small.hip)
hipError_t launch_sample_table(const double* logp, int K, int64_t N, const double* u, uint64_t seed, uint64_t sweep,
                               int64_t row0, int32_t* labels, double* lognorms, hipStream_t stream);
// random column-normalised (K, N) table from the Philox stream (mimo_small.hip)
hipError_t launch_random_resp(double* resp, int K, int64_t N, uint64_t seed, int64_t row0, hipStream_t stream);

// rows with missing values (mimo_small.hip): scan (+ zero the rows and write the mask when `write`), masked labels / tables
// flag |= 1 if any element is a NaN; sums (or null): += the content checksum of mimo_host_checksum (two 64-bit words, zeroed by the caller)
hipError_t launch_nan_any(const double* Z, int64_t count, unsigned int* flag, int num_cu, hipStream_t stream, unsigned long long* sums = nullptr);
hipError_t launch_nan_scan(double* Z, int64_t N, int D, double* mask, unsigned long long* count, bool write, hipStream_t stream);
hipError_t launch_mask_labels(const int32_t* labels, const double* mask, int32_t* out, int64_t N, int K,
                              unsigned long long* bad_counts, hipStream_t stream);
hipError_t launch_mask_table(const double* table, const double* mask, double* out, int K, int64_t N, hipStream_t stream);

hipError_t launch_gather_columns(const double* table, int K, int64_t N, const int64_t* cols, int64_t ncols, double* out, hipStream_t stream);

// shader clock under float64 load (mimo_small.hip): out[2 g] = shader-clock ticks, out[2 g + 1] = 100 MHz ticks of workgroup g
hipError_t launch_clock_probe(unsigned long long* out, int grid, int iters, hipStream_t stream);

// wide shapes (Dz > 16), statistics of a K-major weight table: one 8-wave workgroup per CU (mimo_wide.hip)
bool wide_stats_covers(int K16, int D);
int wide_stats_group_ncb(int K16, int ncb_total);     // feature column blocks per launch
hipError_t launch_wide_stats(const KernelArgs& a, int grid, hipStream_t stream);

// ... and their E-step (softmax table or label draw)
bool wide_estep_covers(int K16, int D, int F16, int gibbs);
hipError_t launch_wide_estep(const KernelArgs& a, int grid, hipStream_t stream);

// label statistics (mimo_rowwave.hip): launches of one pass — 1, or the slice groups of the Dz > 16 / large-K kernel
int label_stats_launches(int K, int D, int structure);
bool label_stats_sorted(int K, int D, int structure);       // the one-pass kernel over the ranked tiles serves the shape (needs the presort buffers)
void set_sorted_range_cap(int tiles);                        // tiles per range of the one-pass kernel at most (mimo_tune "sorted_range"; 0: default)
// ... and the slot-table variant for skewed label vectors (K >= 17, Dz <= 9, N >= 2^17): needs KernelArgs::aux
bool label_stats_uses_slots(int K, int D, int64_t N);
size_t label_stats_aux_words();
hipError_t launch_label_hist_reset(const KernelArgs& a, hipStream_t stream);
bool gibbs_rowwave_counts_labels(int K, int F16, int ZS);

// row-owner softmax + statistics pass, K <= 64, Dz <= 9 (mimo_rowwave.hip); theta in the row-owner image
struct KernelArgs;
bool vi_rowwave_covers(int K, int F16, int ZS);
hipError_t launch_vi_rowwave(const KernelArgs& a, int grid, hipStream_t stream);

// narrow shapes (F <= 16 features, 32 < K <= 128) on v_mfma_f64_4x4x4_4b (mimo_narrow.hip); theta in the narrow image
// [NSF V][16]: slice s V + c, entry 4 k + j = Theta[component j V + c][feature 4 s + k]
int narrow_v(int K);                 // component slots per lane (4 V >= K), 0: not instantiated
int narrow_nsf(int F);               // contraction steps: ceil(F / 4)
// (gibbs: 0 softmax + statistics pass, 1 label draw, 2 label draw + statistics of the labels in the same pass)
bool narrow_covers(int K, int F, int D, int ZS, int gibbs);
void set_narrow_big_vi(int k);      // mimo_tune "narrow_big_vi": largest K of the softmax pass on mimo_narrow_big.hip (0: measured rule)
// the grouped variant (full feature map, Dz = 5 .. 32; rows of the upper triangle padded to whole steps): Dz if it serves the
// shape, else 0; contraction steps of the image either way; position of feature (a, b) in the grouped order
int narrow_dt(int K, int F, int D, int gibbs);
int narrow_steps(int K, int F, int D, int gibbs);
void narrow_group_pos(int D, int a, int b, int* step, int* j);
int narrow_grid(const KernelArgs& a, int num_cu, int F, int gibbs);
hipError_t launch_narrow(const KernelArgs& a, int F, int gibbs, int grid, hipStream_t stream);

// mid shapes (K <= 32 over Dz = 9 .. 32, full map) on row-owner E-step waves + column-owner statistics waves (mimo_mid.hip);
// theta in the grouped image [steps][KB][64] (+ mid_pf() zero slices): slice (s, rb), entry 16 k + i = Theta[16 rb + i][feature
// (a_s, b0_s + k)] of the grouped order (narrow_group_pos)
bool mid_covers(int K, int D, int structure);
int mid_steps(int D);
int mid_pf();
int mid_rows_per_step(int K, int D);
int mid_grid(const KernelArgs& a, int num_cu);
hipError_t launch_mid(const KernelArgs& a, int grid, hipStream_t stream);
// ... and their label pass (K <= 48, Dz = 10 .. 32): E-step + draw on row-owner waves; theta in the grouped image with the components
// permuted as in the row-owner label kernels (slot i of row block rb = component (i & 3) V + 4 rb + (i >> 2), V = 4 KB)
bool mid_labels_covers(int K, int D, int structure);
int mid_labels_grid(const KernelArgs& a, int num_cu);
hipError_t launch_mid_labels(const KernelArgs& a, int grid, hipStream_t stream);

}  // namespace mimo
