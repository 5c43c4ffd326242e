// C-ABI layer of libmimo_hip.so (see include/mimo_hip.h for the contract and the reference
// call sites each entry point replaces).  Host-side work here is O(K D^2): converting the
// canonical (c, b, W) parameters to the feature-space MFMA operand image, and launching /
// sequencing the gfx950 kernels of mimo_kernels.hip on the context's stream.
#include "../../include/mimo_hip.h"
#include "mimo_kernels.h"
#include "mimo_extra.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

using namespace mimo;

namespace mimo_comm {      // mimo_comm.cpp (RCCL through dlopen)
int unique_id(char* out128, char* msg, size_t msglen);
int init(void** comm, const char* id128, int rank, int world, char* msg, size_t msglen);
int destroy(void* comm);
int allreduce_sum_f64(void* comm, double* buf, size_t count, hipStream_t stream, char* msg, size_t msglen);
}

struct mimo_ctx {
  int device = 0;
  int num_cu = 256;       // what the grids are sized from (mimo_tune "num_cu" overrides it for tests)
  int hw_num_cu = 256;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  char err[512] = {0};   // fixed buffer: reporting an error never allocates

  // data
  const double* Z = nullptr;  // device
  double* Z_owned = nullptr;
  int64_t N = 0;
  int D = 0;
  int64_t row0 = 0;

  // feature map for the current D and structure (0: full symmetric W, 1: diagonal W)
  int structure = 0;
  int feat_structure = -1;
  int feat_D = -1;
  int F = 0, F16 = 0;
  std::vector<uint8_t> feat_h;
  uint8_t* feat_d = nullptr;
  uint8_t* feat_full_d = nullptr;   // full map of the current D (small-shape kernel under a structure hint)
  int feat_full_D = -1;

  // parameter image
  double* theta_d = nullptr;  size_t theta_cap = 0;
  double* theta_h = nullptr;  size_t theta_hcap = 0;   // pinned staging

  // workspaces
  double* partials = nullptr; size_t partials_cap = 0;
  double* reduced = nullptr;  size_t reduced_cap = 0;
  double* S_d = nullptr;      size_t S_cap = 0;         // packed stats + 3 scalars
  double* S_h = nullptr;      size_t S_hcap = 0;        // pinned staging

  // optional device-resident tables
  double* resp = nullptr;  size_t resp_cap = 0;  int resp_K = 0;  bool resp_valid = false;
  double* logp = nullptr;  size_t logp_cap = 0;  int logp_K = 0;  bool logp_valid = false;
  double* lse = nullptr;   size_t lse_cap = 0;   bool lse_valid = false;
  int32_t* labels = nullptr; size_t labels_cap = 0; bool labels_valid = false;
  double* u_d = nullptr;   size_t u_cap = 0;
  bool weights_resident = false;     // u_d holds the row weights of the last mimo_estep_weighted (not uniforms of a label pass)
  double* win = nullptr;   size_t win_cap = 0;    // staged host weights
  int32_t* lin = nullptr;  size_t lin_cap = 0;    // staged host labels

  // rows with missing values (NaN): zeroed in the owned copy, excluded from every statistic through the mask
  double* row_mask = nullptr;   size_t mask_cap = 0;      // (N,) 1 = complete row
  int64_t n_bad = 0;
  unsigned long long* cnt_d = nullptr;                    // [1 + 256 + 2]: scan count, labels drawn on NaN rows per component, content checksum of the upload
  uint64_t data_sum[2] = {0, 0};                          // mimo_data_checksum: the rows as mimo_upload received them
  bool data_sum_valid = false;
  int32_t* labels_tmp = nullptr; size_t labels_tmp_cap = 0;
  uint32_t* ls_aux = nullptr;                              // label histogram + slot table of label_stats_slots_kernel
  uint16_t* sort_list = nullptr; size_t sort_list_cap = 0;   // tiles ranked once for a multi-launch label-statistics pass (label_tile_sort_kernel)
  uint16_t* sort_start = nullptr; size_t sort_start_cap = 0;
  double* table_tmp = nullptr;  size_t table_tmp_cap = 0;
  int bad_counts_K = 0;         // > 0: cnt_d[1..K] holds the label counts of the NaN rows of the last label pass

  void* comm = nullptr;         // RCCL communicator (mimo_comm_init): every pass then returns statistics summed over the ranks
  int comm_world = 1;
  bool rowwave_vi_call = false; // set by mimo_estep for the call in progress: row-owner softmax + statistics kernel
  bool rowwave_call = false;    // set by mimo_gibbs_labels for the call in progress: Theta was uploaded in the row-owner layout
  bool mid_labels_call = false; // set for the call in progress: label pass on the mid kernel (Theta in the permuted grouped image)
  bool mid_call = false;        // set for the call in progress: Theta is in the grouped image of the mid kernel (mimo_mid.hip)
  int narrow_call = 0;          // set for the call in progress: Theta is in the narrow image (1: softmax + statistics pass, 2: label pass, 3: label pass + statistics fused)

  // pending asynchronous call (MIMO_F_ASYNC)
  bool pending_async = false;
  size_t pending_slen = 0;
  bool pending_stats = false;

  // profiling: HIP events around every kernel of a pass, on the launch stream
  bool prof = false;
  struct ProfEvent { hipEvent_t e0, e1; int name; };
  std::vector<ProfEvent> pending;
  static constexpr int kProfNames = 16;
  const char* prof_name[kProfNames] = {nullptr};
  double prof_name_ms[kProfNames] = {0.0};
  int64_t prof_name_n[kProfNames] = {0};
  double prof_ms = 0.0;      // all kernels
  int64_t prof_n = 0;        // passes (run_fused calls)
};

static char g_err[512] = {0};

// feature-tile row padding (doubles); MIMO_RS_PAD overrides for bank-conflict experiments
static int rs_pad() {
  static const int v = [] { const char* e = getenv("MIMO_RS_PAD"); return e ? atoi(e) : 1; }();
  return v;
}
// weight-tile row padding (doubles); MIMO_LS_PAD overrides for bank-conflict experiments
// (default 2; 1 for the Dz >= 14, 49 <= K <= 64 shapes of the single-pass kernels — C2: kernel 6.62 -> 6.57 ms, measured with
//  tools/pad_sweep.sh; elsewhere 1 and 2 are within +-2 % of each other, tools/pad_shapes.sh)
static int ls_pad(int K16, int D) {
  static const int v = [] { const char* e = getenv("MIMO_LS_PAD"); return e ? atoi(e) : 0; }();
  return v > 0 ? v : (K16 == 4 && D >= 14 && D <= kMaxFusedD) ? 1 : 2;
}
#ifdef MIMO_STAMPS
static unsigned long long* g_stamps = nullptr;
static int g_stamps_grid = 0;
extern "C" int mimo_debug_stamps_grid() { return g_stamps_grid; }
extern "C" int mimo_debug_stamps_trace(unsigned long long* out128) {   // phase boundaries of workgroups 0 and grid / 2
  return hipMemcpy(out128, g_stamps + (size_t)3 * 8192 * 32, 128 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
static int g_stamps_sel = 0;                       // region: 0 fused / statistics, 1 chunked E-step, 2 wide E-step
extern "C" void mimo_debug_stamps_select(int sel) { g_stamps_sel = sel; }
extern "C" int mimo_debug_stamps(double* out8) {   // mean cycles per wave of each phase, last launch
  std::vector<unsigned long long> h((size_t)g_stamps_grid * 32);
  if (hipMemcpy(h.data(), g_stamps + (size_t)g_stamps_sel * 8192 * 32, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  for (int i = 0; i < 8; ++i) out8[i] = 0;
  for (size_t w = 0; w < h.size() / 8; ++w) for (int i = 0; i < 8; ++i) out8[i] += (double)h[w * 8 + i];
  for (int i = 0; i < 8; ++i) out8[i] /= (double)(h.size() / 8);
  return 0;
}
#endif

static int fail(mimo_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  memcpy(ctx ? ctx->err : g_err, buf, sizeof buf);
  return code;
}

// Every extern "C" entry point runs inside this guard: no C++ exception crosses the boundary (include/mimo_hip.h).
// std::bad_alloc (the std::vector staging buffers, the event list of the profiler) -> MIMO_E_NOMEM, anything else ->
// MIMO_E_INTERNAL; the message goes to the fixed error buffer, so the handlers themselves cannot throw.
template <typename F>
static int guarded(mimo_ctx* ctx, F&& f) noexcept {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    return fail(ctx, MIMO_E_NOMEM, "out of host memory");
  } catch (const std::exception& e) {
    return fail(ctx, MIMO_E_INTERNAL, "internal error: %s", e.what());
  } catch (...) {
    return fail(ctx, MIMO_E_INTERNAL, "internal error: unknown exception");
  }
}

#define HIP_TRY(ctx, expr)                                                                \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(ctx, MIMO_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));        \
  } while (0)

template <typename T>
static int ensure_dev(mimo_ctx* ctx, T** p, size_t* cap, size_t count) {
  if (*cap >= count && *p) return MIMO_OK;
  if (*p) { HIP_TRY(ctx, hipFree(*p)); *p = nullptr; *cap = 0; }
  HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
  *cap = count;
  return MIMO_OK;
}

template <typename T>
static int ensure_pinned(mimo_ctx* ctx, T** p, size_t* cap, size_t count) {
  if (*cap >= count && *p) return MIMO_OK;
  if (*p) { HIP_TRY(ctx, hipHostFree(*p)); *p = nullptr; *cap = 0; }
  HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void**>(p), count * sizeof(T), hipHostMallocDefault));
  *cap = count;
  return MIMO_OK;
}

static int bind(mimo_ctx* ctx) {
  if (!ctx) return fail(nullptr, MIMO_E_INVALID, "null context");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  return MIMO_OK;
}

// feature table for dimension D: pairs (a,b), a <= b <= D over z~ = [z, 1]; padding -> (D+1,D+1)
static int fidx(const mimo_ctx* ctx, int a, int b) {   // feature of the pair (a, b), a <= b <= D; -1: not in the map
  const int D = ctx->D;
  if (ctx->structure == MIMO_STRUCT_DIAG) return (a == b || b == D) ? diag_feat_index(D, a, b) : -1;
  if (ctx->structure == MIMO_STRUCT_LINEAR) return b == D ? a : -1;
  return feat_index(D, a, b);
}

static int prepare_features(mimo_ctx* ctx, int D) {
  if (ctx->feat_D == D && ctx->feat_structure == ctx->structure) return MIMO_OK;
  const int st = ctx->structure;
  ctx->F = st == MIMO_STRUCT_DIAG ? diag_feat_count(D) : st == MIMO_STRUCT_LINEAR ? lin_feat_count(D) : feat_count(D);
  ctx->F16 = (ctx->F + 15) / 16 * 16;
  ctx->feat_h.assign((size_t)ctx->F16 * 2, (uint8_t)(D + 1));
  for (int a = 0; a <= D; ++a)
    for (int b = a; b <= D; ++b) {
      const int f = fidx(ctx, a, b);
      if (f < 0) continue;
      ctx->feat_h[2 * f] = (uint8_t)a;
      ctx->feat_h[2 * f + 1] = (uint8_t)b;
    }
  if (ctx->feat_d) { HIP_TRY(ctx, hipFree(ctx->feat_d)); ctx->feat_d = nullptr; }
  HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->feat_d), ctx->feat_h.size()));
  HIP_TRY(ctx, hipMemcpy(ctx->feat_d, ctx->feat_h.data(), ctx->feat_h.size(), hipMemcpyHostToDevice));
  ctx->feat_D = D;
  ctx->feat_structure = ctx->structure;
  if (D <= kSmallMaxD && ctx->feat_full_D != D) {
    uint8_t full[2 * 16];
    memset(full, D + 1, sizeof full);
    for (int aa = 0; aa <= D; ++aa)
      for (int bb = aa; bb <= D; ++bb) { full[2 * feat_index(D, aa, bb)] = (uint8_t)aa; full[2 * feat_index(D, aa, bb) + 1] = (uint8_t)bb; }
    if (!ctx->feat_full_d) HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->feat_full_d), sizeof full));
    HIP_TRY(ctx, hipMemcpy(ctx->feat_full_d, full, sizeof full, hipMemcpyHostToDevice));
    ctx->feat_full_D = D;
  }
  return MIMO_OK;
}

static int check_shapes(mimo_ctx* ctx, int K) {
  if (ctx->pending_async) return fail(ctx, MIMO_E_STATE, "an asynchronous call is pending: call mimo_wait first");
  if (!ctx->Z) return fail(ctx, MIMO_E_NODATA, "no data uploaded or attached");
  if (K < 1) return fail(ctx, MIMO_E_INVALID, "K must be >= 1 (got %d)", K);
  if (K > 256) return fail(ctx, MIMO_E_UNSUPPORTED, "K = %d > 256 is not covered by the fused kernels", K);
  return MIMO_OK;
}

static void fill_args(mimo_ctx* ctx, int K, KernelArgs* a) {
  memset(a, 0, sizeof *a);
  a->Z = ctx->Z; a->N = ctx->N; a->D = ctx->D; a->K = K; a->K16 = (K + 15) / 16;
  a->F16 = ctx->F16;
  a->ZS = (ctx->D + 2) | 1;      // odd stride: conflict-free row reads
  if (a->K16 > 12) a->ZS = ctx->D + 2;   // K > 192: every byte counts to keep two workgroups per CU (<= 80 KB each)
  a->RS = ctx->F16 + rs_pad();
  a->F16_total = ctx->F16;
  a->cb0 = 0;
  a->write_scalars = 1;
  a->LS = a->K16 * 16 + ls_pad(a->K16, ctx->D);
  a->feat = ctx->feat_d;
  a->row0 = ctx->row0;
  a->do_stats = 1;
  a->diag = ctx->structure != 0;
  a->ntiles = (ctx->N + kTile - 1) / kTile;
  a->aux = ctx->ls_aux;
}

// (c, b, W) -> Theta[k][f] -> MFMA A-operand image [K16][F16/4][64] on the device.
//   f = (D,D): c_k ; (a,D): b_k[a] ; (a,a): -W_aa/2 ; (a,b), a<b: -(W_ab + W_ba)/2
// Does this (data, K) run on the small-shape VALU kernel (mimo_small.hip)?  Dz <= 4, K <= 32, 16-byte aligned rows.
static bool use_small(const mimo_ctx* ctx, int K) {
  static const bool on = [] { const char* e = getenv("MIMO_SMALL"); return !e || atoi(e) != 0; }();   // tuning knob
  return on && small_covers(ctx->D, K) && (reinterpret_cast<uintptr_t>(ctx->Z) % 16) == 0;
}

// small-shape kernel: Theta[G KL][F] row-major over the FULL feature map (feat_index order); a structure hint only
// decides which entries of W are read (diagonal: W_aa; linear: none — the shared quadratic term stays with the caller)
static int upload_theta_small(mimo_ctx* ctx, const double* c, const double* b, const double* W, int K, double* inline_out) {
  const int D = ctx->D, F = feat_count(D), Kp = small_g(D, K) * small_kl(D, K);
  const size_t count = (size_t)Kp * F;
  // one lane per row (G = 1): Theta goes into the kernel arguments — no staging buffer to wait for, no transfer to enqueue
  const bool inl = inline_out && small_g(D, K) == 1 && count <= (size_t)kThetaInline;
  int rc;
  if (!inl) {
    if ((rc = ensure_dev(ctx, &ctx->theta_d, &ctx->theta_cap, count))) return rc;
    if ((rc = ensure_pinned(ctx, &ctx->theta_h, &ctx->theta_hcap, count))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));    // the staging buffer may still be in flight
  }
  double* img = inl ? inline_out : ctx->theta_h;
  memset(img, 0, count * sizeof(double));
  bool finite = true;
  auto chk = [&](double v) { finite = finite && std::fabs(v) <= 1.7976931348623157e308; return v; };
  for (int k = 0; k < K; ++k) {
    double* t = img + (size_t)k * F;
    const double* bk = b + (size_t)k * D;
    const double* Wk = W + (size_t)k * D * D;
    if (c[k] != c[k] || c[k] > 1.7976931348623157e308) return fail(ctx, MIMO_E_INVALID, "c[%d] is NaN or +inf", k);
    t[F - 1] = c[k] < kPadLogDensity ? kPadLogDensity : c[k];
    for (int a = 0; a < D; ++a) t[feat_index(D, a, D)] = chk(bk[a]);
    if (ctx->structure == MIMO_STRUCT_LINEAR) {
      if (k > 0 && memcmp(Wk, W, sizeof(double) * D * D) != 0)
        return fail(ctx, MIMO_E_INVALID, "linear structure is set (mimo_set_structure) but W[%d] differs from W[0]", k);
      continue;
    }
    for (int a = 0; a < D; ++a) {
      t[feat_index(D, a, a)] = chk(-0.5 * Wk[a * D + a]);
      for (int bb = a + 1; bb < D; ++bb) {
        if (ctx->structure == MIMO_STRUCT_FULL) t[feat_index(D, a, bb)] = chk(-0.5 * (Wk[a * D + bb] + Wk[bb * D + a]));
        else if (Wk[a * D + bb] != 0.0 || Wk[bb * D + a] != 0.0)
          return fail(ctx, MIMO_E_INVALID, "diagonal structure is set (mimo_set_structure) but W[%d] has the "
                      "off-diagonal entry (%d,%d)", k, a, bb);
      }
    }
  }
  if (!finite) return fail(ctx, MIMO_E_INVALID, "b or W holds a NaN or an infinity");
  for (int k = K; k < Kp; ++k) img[(size_t)k * F + F - 1] = kPadLogDensity;
  if (!inl) HIP_TRY(ctx, hipMemcpyAsync(ctx->theta_d, img, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  return MIMO_OK;
}

// Label pass on the row-owner kernels (mimo_rowwave.hip): Dz <= 9 (beyond the small-shape kernel's range), full structure, nothing but labels
// (+ their statistics) requested.
static bool use_rowwave(const mimo_ctx* ctx, int K, bool wants_tables) {
  static const bool on = [] { const char* e = getenv("MIMO_ROWWAVE"); return !e || atoi(e) != 0; }();   // tuning knob
  if (!on || wants_tables) return false;
  const int ZS = (K + 15) / 16 > 12 ? ctx->D + 2 : ((ctx->D + 2) | 1);      // as fill_args
  return rowwave_covers(K, ctx->F16, ZS) && label_stats_covers(K, ctx->D, ctx->structure);
}

// Theta image of the row-owner label kernel: [NS][KB][64]; component k sits in A-row (k / V) + 4 (k % 4) of row block
// (k % V) / 4, V = 4 KB, so that an output lane holds a contiguous quarter of the components (gibbs_rowwave_kernel)
static int upload_theta_rowwave(mimo_ctx* ctx, const double* c, const double* b, const double* W, int K) {
  const int D = ctx->D;
  const int ZSk = (K + 15) / 16 > 12 ? D + 2 : ((D + 2) | 1);      // as fill_args
  const int KB = rowwave_kb_shape(K, ctx->F16, ZSk), V = 4 * KB;
  const int NS = rowwave_image_ns(K, ctx->F16, ZSk);                // (whole chunks where the label kernel streams Theta)
  const size_t count = (size_t)NS * KB * 64;
  int rc;
  if ((rc = ensure_dev(ctx, &ctx->theta_d, &ctx->theta_cap, count))) return rc;
  if ((rc = ensure_pinned(ctx, &ctx->theta_h, &ctx->theta_hcap, count))) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  double* img = ctx->theta_h;
  memset(img, 0, count * sizeof(double));
  bool finite = true;
  auto put = [&](int k, int f, double v) {
    const int t = k % V, rb = t / 4, i = k / V + 4 * (t % 4);
    finite = finite && std::fabs(v) <= 1.7976931348623157e308;
    img[((size_t)(f / 4) * KB + rb) * 64 + (f % 4) * 16 + i] = v;
  };
  for (int k = 0; k < K; ++k) {
    const double* bk = b + (size_t)k * D;
    const double* Wk = W + (size_t)k * D * D;
    if (c[k] != c[k] || c[k] > 1.7976931348623157e308) return fail(ctx, MIMO_E_INVALID, "c[%d] is NaN or +inf", k);
    put(k, fidx(ctx, D, D), c[k] < kPadLogDensity ? kPadLogDensity : c[k]);
    for (int a = 0; a < D; ++a) put(k, fidx(ctx, a, D), bk[a]);
    if (ctx->structure == MIMO_STRUCT_LINEAR) {       // the shared quadratic term stays with the caller (mimo_set_structure)
      if (k > 0 && memcmp(Wk, W, sizeof(double) * D * D) != 0)
        return fail(ctx, MIMO_E_INVALID, "linear structure is set (mimo_set_structure) but W[%d] differs from W[0]", k);
      continue;
    }
    for (int a = 0; a < D; ++a) {
      put(k, fidx(ctx, a, a), -0.5 * Wk[a * D + a]);
      for (int bb = a + 1; bb < D; ++bb) {
        if (ctx->structure == MIMO_STRUCT_FULL) put(k, feat_index(D, a, bb), -0.5 * (Wk[a * D + bb] + Wk[bb * D + a]));
        else if (Wk[a * D + bb] != 0.0 || Wk[bb * D + a] != 0.0)
          return fail(ctx, MIMO_E_INVALID, "diagonal structure is set (mimo_set_structure) but W[%d] has the "
                      "off-diagonal entry (%d,%d)", k, a, bb);
      }
    }
  }
  if (!finite) return fail(ctx, MIMO_E_INVALID, "b or W holds a NaN or an infinity");
  for (int k = K; k < 16 * KB; ++k) put(k, fidx(ctx, D, D), kPadLogDensity);
  HIP_TRY(ctx, hipMemcpyAsync(ctx->theta_d, img, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  return MIMO_OK;
}

// Passes of the narrow shapes (mimo_narrow.hip: F <= 16 features, 32 < K <= 128 on the 4x4x4 matrix instruction): plain
// requests only — nothing but statistics + scalars (softmax pass) or labels + their statistics (label pass).
// Returns the kernel mode + 1 (1: softmax + statistics, 2: label draw with the label-statistics kernel behind it, 3: label draw +
// statistics in one pass — few components over many features, MIMO_NARROW_FUSED_LABELS=0: off) or 0.
static int use_narrow(const mimo_ctx* ctx, int K, bool gibbs, bool plain, bool stats) {
  static const bool fused_labels = [] { const char* e = getenv("MIMO_NARROW_FUSED_LABELS"); return !e || atoi(e) != 0; }();
  if (!plain) return 0;                 // (the small-shape kernel keeps the generic requests of its range and the shapes below narrow_covers' K)
  const int ZS = (K + 15) / 16 > 12 ? ctx->D + 2 : ((ctx->D + 2) | 1);      // as fill_args
  if (!gibbs) return narrow_covers(K, ctx->F, ctx->D, ZS, 0) ? 1 : 0;     // (rows with NaN: their mask is the row-weight vector of the pass)
  const bool two = narrow_covers(K, ctx->F, ctx->D, ZS, 1) && label_stats_covers(K, ctx->D, ctx->structure);
  const bool one = fused_labels && ctx->n_bad == 0 && narrow_covers(K, ctx->F, ctx->D, ZS, 2);
  if (one && !two) return 3;
  // both exist: the fused pass wins while its second product is cheap next to a second pass over Z (profiles/r03_wide_sweep_fused_labels.txt,
  // N = 2e6, us per sweep, label kernel + label statistics / fused: Dz=8 K=4 154 / 93, K=8 154 / 129, K=16 181 / 194; Dz=12 K=8 237 / 208,
  // K=16 312 / 395; Dz=16 K=4 256 / 213, K=8 315 / 374, K=16 430 / 649)
  const int V = narrow_v(K), D = ctx->D;
  if (one && stats && (V == 1 || (D <= 6 && V <= 6) || (D <= 12 && V <= 3))) return 3;
  return two ? 2 : 0;
}

// Theta image of the narrow kernels: [NSF][V][16]; slice s V + c, entry 4 kk + j = Theta[component j V + c][feature 4 s + kk]
// (an output lane holds a contiguous quarter of the components: narrow_kernel)
// Grouped variant (narrow_dt): the steps follow the rows of the upper triangle, feature (a, b) at narrow_group_pos.
static int upload_theta_narrow(mimo_ctx* ctx, const double* c, const double* b, const double* W, int K, int mode) {
  const int D = ctx->D, dt = narrow_dt(K, ctx->F, D, mode), NSF = narrow_steps(K, ctx->F, D, mode), V = narrow_v(K);
  std::vector<int> gpos;                // grouped: 4 step + index of every feature of the full map
  if (dt) {
    gpos.assign((size_t)ctx->F, 0);
    for (int aa = 0; aa <= D; ++aa)
      for (int bb = aa; bb <= D; ++bb) {
        int st, j;
        narrow_group_pos(D, aa, bb, &st, &j);
        gpos[feat_index(D, aa, bb)] = 4 * st + j;
      }
  }
  const size_t count = (size_t)NSF * V * 16;
  int rc;
  if ((rc = ensure_dev(ctx, &ctx->theta_d, &ctx->theta_cap, count))) return rc;
  if ((rc = ensure_pinned(ctx, &ctx->theta_h, &ctx->theta_hcap, count))) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  double* img = ctx->theta_h;
  memset(img, 0, count * sizeof(double));
  bool finite = true;
  auto put = [&](int k, int f, double v) {
    finite = finite && std::fabs(v) <= 1.7976931348623157e308;
    const int g = dt ? gpos[f] : f;
    img[((size_t)(g / 4) * V + k % V) * 16 + 4 * (g % 4) + k / V] = v;
  };
  for (int k = 0; k < K; ++k) {
    const double* bk = b + (size_t)k * D;
    const double* Wk = W + (size_t)k * D * D;
    if (c[k] != c[k] || c[k] > 1.7976931348623157e308) return fail(ctx, MIMO_E_INVALID, "c[%d] is NaN or +inf", k);
    put(k, fidx(ctx, D, D), c[k] < kPadLogDensity ? kPadLogDensity : c[k]);
    for (int a = 0; a < D; ++a) put(k, fidx(ctx, a, D), bk[a]);
    if (ctx->structure == MIMO_STRUCT_LINEAR) {       // the shared quadratic term stays with the caller (mimo_set_structure)
      if (k > 0 && memcmp(Wk, W, sizeof(double) * D * D) != 0)
        return fail(ctx, MIMO_E_INVALID, "linear structure is set (mimo_set_structure) but W[%d] differs from W[0]", k);
      continue;
    }
    for (int a = 0; a < D; ++a) {
      put(k, fidx(ctx, a, a), -0.5 * Wk[a * D + a]);
      for (int bb = a + 1; bb < D; ++bb) {
        if (ctx->structure == MIMO_STRUCT_FULL) put(k, feat_index(D, a, bb), -0.5 * (Wk[a * D + bb] + Wk[bb * D + a]));
        else if (Wk[a * D + bb] != 0.0 || Wk[bb * D + a] != 0.0)
          return fail(ctx, MIMO_E_INVALID, "diagonal structure is set (mimo_set_structure) but W[%d] has the "
                      "off-diagonal entry (%d,%d)", k, a, bb);
      }
    }
  }
  if (!finite) return fail(ctx, MIMO_E_INVALID, "b or W holds a NaN or an infinity");
  for (int k = K; k < 4 * V; ++k) put(k, fidx(ctx, D, D), kPadLogDensity);
  HIP_TRY(ctx, hipMemcpyAsync(ctx->theta_d, img, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  return MIMO_OK;
}

// Softmax + statistics pass of the mid shapes (mimo_mid.hip): plain requests (statistics + scalars, row weights / the NaN mask
// allowed), full feature map, K <= 32 where neither the narrow kernels (few components) nor the single-pass tile kernels do
// better — measured per shape (profiles/r04_mid_kernel_sweep.txt): from Dz = 17 everything the narrow kernels do not take;
// MIMO_MID_MIN_D moves the lower end (tuning knob)
static int g_mid_min_d = [] { const char* e = getenv("MIMO_MID_MIN_D"); return e ? atoi(e) : 0; }();       // (mimo_tune "mid_min_d"; 0: the measured rule)
static int g_mid_narrow_k = [] { const char* e = getenv("MIMO_MID_NARROW_K"); return e ? atoi(e) : 0; }(); // (mimo_tune "mid_narrow_k")
static int use_narrow(const mimo_ctx* ctx, int K, bool gibbs, bool plain, bool stats);
static bool use_mid(const mimo_ctx* ctx, int K, bool plain) {
  const int D = ctx->D;
  if (!plain || !mid_covers(K, D, ctx->structure)) return false;
  if (g_mid_min_d > 0 || g_mid_narrow_k > 0)       // forced by the caller (tests, sweeps)
    return D >= (g_mid_min_d > 0 ? g_mid_min_d : 5) &&
           (K >= (g_mid_narrow_k > 0 ? g_mid_narrow_k : 33) || !use_narrow(ctx, K, false, plain, true));
  // measured (tools/mid_sweep.py, profiles/r04_mid_kernel_sweep.txt; fraction of the float64 rate, other route -> mid):
  //   K = 17 .. 32: from Dz = 13 (Dz=13 K=32 0.55 -> 0.61, Dz=14 0.58 -> 0.66, Dz=16 0.61 -> 0.64, Dz=20 0.41 -> 0.69, Dz=32 0.48 -> 0.75;
  //                 Dz=12 K=32 0.61 -> 0.57 and Dz=11 0.53 -> 0.50 stay on the tile / row-owner kernels)
  //   K = 13 .. 16: from Dz = 12 against the narrow kernels (Dz=12 K=16 0.41 -> 0.49, Dz=14 0.44 -> 0.58, Dz=16 0.42 -> 0.57; Dz=11 0.45 -> 0.43)
  //   K <= 12: the narrow kernels where they exist (Dz=16 K=12 0.44 = 0.44, Dz=15 0.42 -> 0.38, Dz=20 K=8 0.37 -> 0.30) up to Dz = 23
  //            (Dz=24 K=8 0.31 -> 0.32, Dz=26 K=8 0.19 -> 0.34); beyond them the two-stage path was all there was (Dz=20 K=12 0.18 -> 0.47)
  //   K = 33 .. 48: from Dz = 9 (Dz=9 K=48 0.48 -> 0.55, Dz=12 0.56 -> 0.60, Dz=16 0.57 -> 0.67, Dz=20 0.46 -> 0.71, Dz=26 0.52 -> 0.80; Dz=10, 11: level)
  //   K = 49 .. 64: from Dz = 18 (Dz=18 K=64 0.64 -> 0.68, Dz=20 0.61 -> 0.76, Dz=21 0.66 -> 0.79, one wave per SIMD from Dz = 22: Dz=24 0.57 -> 0.63,
  //                 Dz=28 0.63 -> 0.68; below, ten column blocks do not divide over eight waves and the tile kernels keep K = 64: Dz=16 0.78 against 0.57)
  //   K = 65 .. 96: wherever the kernels exist from Dz = 6 (five / six row blocks instead of the eight the tile and two-stage kernels pay for:
  //                 Dz=8 K=96 0.43 -> 0.56, Dz=9 K=72 0.35 -> 0.56, Dz=12 K=96 0.46 -> 0.62, Dz=14 K=80 0.35 -> 0.71, Dz=16 K=80 0.42 -> 0.60; with one
  //                 wave per SIMD: Dz=16 K=96 0.47 -> 0.61, Dz=20 K=96 0.48 -> 0.66, Dz=23 K=96 0.51 -> 0.66, Dz=26 K=80 0.49 -> 0.66, Dz=28 K=48 0.47 -> 0.66)
  //   K = 97 .. 128 (seven / eight row blocks, one wave per SIMD): K <= 112 from Dz = 8 (Dz=8 K=112 0.49 -> 0.58, Dz=12 0.53 -> 0.62, Dz=20 0.58 -> 0.65);
  //                 K = 113 .. 128 at Dz = 10 .. 15 (Dz=10 0.43 -> 0.51, Dz=14 0.54 -> 0.64; Dz <= 8: the tile kernels, 0.56 against 0.40; Dz >= 16:
  //                 the wide two-stage kernels are level or ahead, Dz=18 0.74 against 0.70)
  if (K >= 113) return D >= 10 && D <= 15;
  if (K >= 97) return D >= 8;
  if (K >= 65) return D >= 6;
  if (K >= 49) return D >= 18;
  if (K >= 33) return D >= 9;
  if (K >= 17) return D >= 13;
  if (K >= 13) return D >= 12;
  if (K <= 4) return !use_narrow(ctx, K, false, plain, true);        // (one slot of the narrow kernels: Dz=28 K=4 0.30 against 0.17 on 16-padded tiles)
  return D >= 24 || !use_narrow(ctx, K, false, plain, true);
}

// Label pass of the mid shapes (mimo_mid.hip, label mode + the label-statistics kernels): K <= 48 at Dz >= 10, plain requests, where it
// measured ahead of the row-owner label kernels (profiles/r04_mid_label_sweep.txt); "mid_labels_min_d" (mimo_tune) moves the lower end
static int g_mid_labels_min_d = [] { const char* e = getenv("MIMO_MID_LABELS_MIN_D"); return e ? atoi(e) : 0; }();
static int g_bound_promote_mid_k = 16;     // largest K of a mid-kernel shape whose bound-only pass runs as the plain pass (profiles/r04_bound_pass.txt: N = 2e6, ms generic / plain:
                                           // Dz=20 K=16 1.02 / 0.73, Dz=32 K=16 1.73 / 1.51 — but Dz=24 K=32 1.51 / 1.68, Dz=16 K=48 0.99 / 1.31; every narrow shape gains: Dz=2 K=50
                                           // 0.37 / 0.16, Dz=1 K=100 0.77 / 0.20, Dz=4 K=128 0.75 / 0.62, Dz=16 K=4 0.65 / 0.23, Dz=32 K=4 1.73 / 0.81)
static int g_mid_labels_narrow_k = 0;      // (mimo_tune "mid_labels_narrow_k": K from which the label mode goes before the narrow label kernels; 0: measured rule)
static bool use_mid_labels(const mimo_ctx* ctx, int K, bool wants_tables) {
  if (wants_tables || !mid_labels_covers(K, ctx->D, ctx->structure) || !label_stats_covers(K, ctx->D, ctx->structure)) return false;
  if (g_mid_labels_min_d > 0) return ctx->D >= g_mid_labels_min_d;
  // measured (tools/mid_label_sweep.py, N = 2e6, fraction of the float64 rate of the whole sweep, row-owner label kernels -> mid label mode):
  //   K <= 16 from Dz = 17 (the streamed kernel pads to 32 components: Dz=17 K=16 0.23 -> 0.34, Dz=24 0.28 -> 0.45, Dz=32 0.29 -> 0.47; Dz=28 K=8 0.15 -> 0.26)
  //   K = 33 .. 48 from Dz = 14 (Dz=14 0.45 -> 0.49, Dz=20 0.46 -> 0.55, Dz=28 0.48 -> 0.64); K = 17 .. 32 from Dz = 20 (0.48 -> 0.51, Dz=32 0.56 -> 0.58)
  //   below: the row-owner kernels with Theta resident in LDS stay ahead (Dz=16 K=32 0.51 against 0.41)
  const int D = ctx->D;
  return K >= 33 ? D >= 14 : K >= 17 ? D >= 20 : D >= 17;
}

static bool mid_labels_before_narrow(const mimo_ctx* ctx, int K) {
  if (g_mid_labels_narrow_k > 0) return K >= g_mid_labels_narrow_k;
  // measured (profiles/r04_mid_label_sweep.txt, second block): the narrow label kernels stay ahead up to Dz = 20 (Dz=16 K=16 0.34 against 0.33,
  // K=8 0.23 against 0.17); from Dz = 24 their one-wave-per-SIMD variants fall behind for K = 5 .. 8 (Dz=24 K=8 1.11 -> 0.67 ms, Dz=26 1.29 -> 0.74)
  return K >= 5 && ctx->D >= 24;
}

// Theta image of the mid kernel: [steps][KB][64] in the grouped feature order + mid_pf() zero slices
static int upload_theta_mid(mimo_ctx* ctx, const double* c, const double* b, const double* W, int K, bool labels = false) {
  const int D = ctx->D, KB = (K + 15) / 16, NS = mid_steps(D), V = 4 * KB;
  const size_t count = ((size_t)NS * KB + mid_pf()) * 64;
  int rc;
  if ((rc = ensure_dev(ctx, &ctx->theta_d, &ctx->theta_cap, count))) return rc;
  if ((rc = ensure_pinned(ctx, &ctx->theta_h, &ctx->theta_hcap, count))) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  double* img = ctx->theta_h;
  memset(img, 0, count * sizeof(double));
  bool finite = true;
  auto put = [&](int k, int aa, int bb, double v) {
    int st, jj;
    narrow_group_pos(D, aa, bb, &st, &jj);
    finite = finite && std::fabs(v) <= 1.7976931348623157e308;
    int rb = k / 16, i = k % 16;
    if (labels) {                       // label pass: lane quarter q holds components q V .. q V + V - 1 (slot i of row block rb: q = i & 3, r = i >> 2)
      const int qq = k / V, t = k % V;
      rb = t / 4; i = 4 * (t % 4) + qq;
    }
    img[((size_t)st * KB + rb) * 64 + 16 * jj + i] = v;
  };
  for (int k = 0; k < K; ++k) {
    const double* bk = b + (size_t)k * D;
    const double* Wk = W + (size_t)k * D * D;
    if (c[k] != c[k] || c[k] > 1.7976931348623157e308) return fail(ctx, MIMO_E_INVALID, "c[%d] is NaN or +inf", k);
    put(k, D, D, c[k] < kPadLogDensity ? kPadLogDensity : c[k]);
    for (int a = 0; a < D; ++a) {
      put(k, a, D, bk[a]);
      put(k, a, a, -0.5 * Wk[a * D + a]);
      for (int bb = a + 1; bb < D; ++bb) put(k, a, bb, -0.5 * (Wk[a * D + bb] + Wk[bb * D + a]));
    }
  }
  if (!finite) return fail(ctx, MIMO_E_INVALID, "b or W holds a NaN or an infinity");
  for (int k = K; k < 16 * KB; ++k) put(k, D, D, kPadLogDensity);
  HIP_TRY(ctx, hipMemcpyAsync(ctx->theta_d, img, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  return MIMO_OK;
}

static int upload_theta(mimo_ctx* ctx, const double* c, const double* b, const double* W, int K, KernelArgs* a = nullptr) {
  if (use_small(ctx, K)) return upload_theta_small(ctx, c, b, W, K, a ? a->theta_inline : nullptr);
  const int D = ctx->D, F16 = ctx->F16;
  const int K16 = ((K + 15) / 16 <= 4) ? 4 : 16;   // every wave streams 1 (K<=64) or up to 4 row blocks; unused ones are zero
  // fused kernels step through F16/4 slices per row block; the chunked E-step through whole chunks
  const int NS = fused_covers((K + 15) / 16, F16 / 16, kSrcEstep) ? F16 / 4 : chunked_ns_pad(F16);
  const size_t count = (size_t)K16 * NS * 64;
  int rc;
  if ((rc = ensure_dev(ctx, &ctx->theta_d, &ctx->theta_cap, count))) return rc;
  if ((rc = ensure_pinned(ctx, &ctx->theta_h, &ctx->theta_hcap, count))) return rc;
  // the staging buffer may still be in flight from the previous call on this stream
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  double* img = ctx->theta_h;
  memset(img, 0, count * sizeof(double));
  bool finite = true;
  for (int k = 0; k < K; ++k) {
    const int rb = k / 16, i = k % 16;
    const double* bk = b + (size_t)k * D;
    const double* Wk = W + (size_t)k * D * D;
    auto put = [&](int f, double v) {
      const int s = f / 4, kk = f % 4;
      finite = finite && std::fabs(v) <= 1.7976931348623157e308;
      img[((size_t)rb * NS + s) * 64 + kk * 16 + i] = v;
    };
    // a component switched off by its weight (log 0 = -inf in c_k, gmm.py:84 of the host mirror) enters like a
    // padding component: l = -1e300 for every datum, r = 0 — an infinite operand would turn the zero features of the
    // rows past N into NaN statistics
    if (c[k] != c[k] || c[k] > 1.7976931348623157e308)
      return fail(ctx, MIMO_E_INVALID, "c[%d] is NaN or +inf", k);
    put(fidx(ctx, D, D), c[k] < kPadLogDensity ? kPadLogDensity : c[k]);
    for (int a = 0; a < D; ++a) put(fidx(ctx, a, D), bk[a]);
    if (ctx->structure == MIMO_STRUCT_LINEAR) {
      // the common quadratic term stays with the caller (see mimo_set_structure); all W[k] must be one matrix
      if (k > 0 && memcmp(Wk, W, sizeof(double) * D * D) != 0)
        return fail(ctx, MIMO_E_INVALID, "linear structure is set (mimo_set_structure) but W[%d] differs from W[0]", k);
      continue;
    }
    for (int a = 0; a < D; ++a) {
      put(fidx(ctx, a, a), -0.5 * Wk[a * D + a]);
      for (int bb = a + 1; bb < D; ++bb) {
        if (ctx->structure == MIMO_STRUCT_FULL) put(feat_index(D, a, bb), -0.5 * (Wk[a * D + bb] + Wk[bb * D + a]));
        else if (Wk[a * D + bb] != 0.0 || Wk[bb * D + a] != 0.0)
          return fail(ctx, MIMO_E_INVALID, "diagonal structure is set (mimo_set_structure) but W[%d] has the "
                      "off-diagonal entry (%d,%d)", k, a, bb);
      }
    }
  }
  if (!finite) return fail(ctx, MIMO_E_INVALID, "b or W holds a NaN or an infinity");
  // padding components of the last row block: l = -1e300 for every datum, so the normalise phase needs no
  // "does this component exist" test (exp -> 0, never the maximum, zero weight in the statistics)
  for (int k = K; k < 16 * ((K + 15) / 16); ++k) {
    const int f = fidx(ctx, D, D);
    img[((size_t)(k / 16) * NS + f / 4) * 64 + (f % 4) * 16 + k % 16] = kPadLogDensity;
  }
  HIP_TRY(ctx, hipMemcpyAsync(ctx->theta_d, img, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  return MIMO_OK;
}

static void drain_profile(mimo_ctx* ctx) {
  for (auto& pr : ctx->pending) {
    float ms = 0.f;
    if (hipEventSynchronize(pr.e1) == hipSuccess && hipEventElapsedTime(&ms, pr.e0, pr.e1) == hipSuccess) {
      ctx->prof_ms += ms;
      ctx->prof_name_ms[pr.name] += ms;
      ctx->prof_name_n[pr.name] += 1;
    }
    (void)hipEventDestroy(pr.e0);
    (void)hipEventDestroy(pr.e1);
  }
  ctx->pending.clear();
}

static int prof_slot(mimo_ctx* ctx, const char* name) {
  for (int i = 0; i < mimo_ctx::kProfNames; ++i) {
    if (ctx->prof_name[i] == name) return i;
    if (!ctx->prof_name[i]) { ctx->prof_name[i] = name; return i; }
  }
  return mimo_ctx::kProfNames - 1;
}

// launch() bracketed by two events on the context's stream when profiling is on
template <typename L>
static int timed_launch(mimo_ctx* ctx, const char* name, L&& launch) {
  if (!ctx->prof) return launch();
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HIP_TRY(ctx, hipEventCreate(&e0));
  if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return fail(ctx, MIMO_E_HIP, "hipEventCreate failed"); }
  (void)hipEventRecord(e0, ctx->stream);
  const int rc = launch();
  (void)hipEventRecord(e1, ctx->stream);
  if (rc != MIMO_OK) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return rc; }
  ctx->pending.push_back({e0, e1, prof_slot(ctx, name)});
  return MIMO_OK;
}

// buffers of the presorted tiles for a label-statistics pass of several launches (Dz >= 10: windows / feature slices)
static int prepare_label_presort(mimo_ctx* ctx, KernelArgs& a) {
  a.sort_list = nullptr; a.sort_start = nullptr;
  if (ctx->structure != 0 || a.D < 10 || a.N < 1 ||
      (label_stats_launches(a.K, a.D, ctx->structure) < 2 && !label_stats_sorted(a.K, a.D, ctx->structure))) return MIMO_OK;
  const size_t tiles128 = (size_t)((a.N + 127) / 128);
  int rc;
  if ((rc = ensure_dev(ctx, &ctx->sort_list, &ctx->sort_list_cap, tiles128 * 128 + 256))) return rc;
  if ((rc = ensure_dev(ctx, &ctx->sort_start, &ctx->sort_start_cap, tiles128 * 257 + 257))) return rc;
  a.sort_list = ctx->sort_list; a.sort_start = ctx->sort_start;
  return MIMO_OK;
}

// run the pass (one fused kernel, the two-stage sequence, or the small-shape kernel) -> reduce -> unpack;
// deliver S / scalars to host or device pointers
static int run_pass(mimo_ctx* ctx, KernelArgs& a, int src, int flags, double* S, double* scalars) {
  const int K = a.K, D = a.D;
  const int Kpad = a.K16 * 16;
  const bool small = use_small(ctx, K) && ctx->narrow_call == 0;
  if (small) { a.F16_total = 16; a.F16 = 16; }
  const bool rowwave = src == kSrcEstep && ctx->rowwave_call;                       // label pass + label statistics
  const bool rowvi = src == kSrcEstep && ctx->rowwave_vi_call;                      // row-owner softmax + statistics pass
  const bool narrow_g1 = src == kSrcEstep && ctx->narrow_call == 3;                 // narrow label pass with the statistics of the labels in the same kernel
  const bool narrow_vi = src == kSrcEstep && (ctx->narrow_call == 1 || narrow_g1);  // narrow softmax + statistics pass (or the above: same launch shape)
  const bool narrow_g = src == kSrcEstep && ctx->narrow_call == 2;                  // narrow label pass + label statistics
  const bool mid_g = src == kSrcEstep && ctx->mid_labels_call;                       // mid label pass + label statistics
  const bool lstats = !small && ((src == kSrcLabels && label_stats_covers(K, D, ctx->structure)) || rowwave || narrow_g || mid_g);
  const bool mid = src == kSrcEstep && ctx->mid_call;                               // mid shapes: row-owner E-step + column-owner statistics
  int grid = small ? small_grid(a, ctx->num_cu, src) : lstats ? label_stats_grid(a, ctx->num_cu)
             : rowvi ? rowwave_grid(a, ctx->num_cu) : narrow_vi ? narrow_grid(a, ctx->num_cu, ctx->F, narrow_g1 ? 2 : 0)
             : mid ? mid_grid(a, ctx->num_cu) : fused_grid(a, ctx->num_cu, src);
  // two-stage pass on the pipelined E-step (mimo_wide.hip): that kernel is built for two workgroups per CU whatever K is
  // (fused_grid's fallback assumes one for K > 128); the statistics launches of the pass share the grid (partial blocks)
  if (!small && !lstats && !rowvi && !narrow_vi && !mid && src == kSrcEstep && !fused_covers(a.K16, a.F16 / 16, src) &&
      wide_estep_covers(a.K16, D, a.F16, a.gibbs)) {
    const int64_t g2 = 2 * (int64_t)ctx->num_cu;
    grid = (int)(g2 < a.ntiles ? g2 : (a.ntiles > 0 ? a.ntiles : 1));
  }
  const size_t pstride = (size_t)Kpad * a.F16 + 4;
  int rc;
  if ((rc = ensure_dev(ctx, &ctx->partials, &ctx->partials_cap, pstride * (size_t)grid))) return rc;
  a.partials = ctx->partials;

#ifdef MIMO_STAMPS
  {
    static unsigned long long* stamps_d = nullptr;
    if (!stamps_d) (void)hipMalloc(reinterpret_cast<void**>(&stamps_d), 4 * 8192 * 4 * 8 * sizeof(unsigned long long));
    a.stamps = stamps_d;
    g_stamps = stamps_d; g_stamps_grid = grid;
  }
#endif
  const int ncb_total = a.F16 / 16;
  if (lstats) {
    if (rowwave) {
      // the resident-Theta label kernel counts its labels for the slot table of the statistics kernel behind it
      // (not with NaN rows: their labels are masked before the statistics; MIMO_FUSE_LABEL_HIST=0: off)
      static const bool fuse_on = [] { const char* e = getenv("MIMO_FUSE_LABEL_HIST"); return !e || atoi(e) != 0; }();
      a.fuse_hist = fuse_on && a.do_stats && ctx->n_bad == 0 && a.aux && label_stats_uses_slots(K, D, a.N) &&
                    gibbs_rowwave_counts_labels(K, a.F16, a.ZS) ? 1 : 0;
      rc = timed_launch(ctx, "gibbs_rowwave_kernel", [&]() -> int {
        if (a.fuse_hist) HIP_TRY(ctx, launch_label_hist_reset(a, ctx->stream));
        HIP_TRY(ctx, launch_gibbs_rowwave(a, rowwave_grid(a, ctx->num_cu), ctx->stream));
        return MIMO_OK;
      });
      if (rc) return rc;
    } else if (mid_g) {
      rc = timed_launch(ctx, "mid_kernel (labels)", [&]() -> int {
        HIP_TRY(ctx, launch_mid_labels(a, mid_labels_grid(a, ctx->num_cu), ctx->stream));
        return MIMO_OK;
      });
      if (rc) return rc;
    } else if (narrow_g) {
      static const bool fuse_on = [] { const char* e = getenv("MIMO_FUSE_LABEL_HIST"); return !e || atoi(e) != 0; }();
      a.fuse_hist = fuse_on && a.do_stats && ctx->n_bad == 0 && a.aux && label_stats_uses_slots(K, D, a.N) ? 1 : 0;
      rc = timed_launch(ctx, "narrow_kernel", [&]() -> int {
        if (a.fuse_hist) HIP_TRY(ctx, launch_label_hist_reset(a, ctx->stream));
        HIP_TRY(ctx, launch_narrow(a, ctx->F, 1, narrow_grid(a, ctx->num_cu, ctx->F, 1), ctx->stream));
        return MIMO_OK;
      });
      if (rc) return rc;
    }
    if (a.do_stats) {
      if ((rc = prepare_label_presort(ctx, a))) return rc;
      rc = timed_launch(ctx, "label_stats_kernel", [&]() -> int {
        HIP_TRY(ctx, launch_label_stats(a, ctx->structure, grid, ctx->stream));
        return MIMO_OK;
      });
      if (rc) return rc;
    }
  } else if (rowvi) {
    rc = timed_launch(ctx, "vi_rowwave_kernel", [&]() -> int {
      HIP_TRY(ctx, launch_vi_rowwave(a, grid, ctx->stream));
      return MIMO_OK;
    });
    if (rc) return rc;
  } else if (narrow_vi) {
    rc = timed_launch(ctx, "narrow_kernel", [&]() -> int {
      HIP_TRY(ctx, launch_narrow(a, ctx->F, narrow_g1 ? 2 : 0, grid, ctx->stream));
      return MIMO_OK;
    });
    if (rc) return rc;
  } else if (mid) {
    rc = timed_launch(ctx, "mid_kernel", [&]() -> int {
      HIP_TRY(ctx, launch_mid(a, grid, ctx->stream));
      return MIMO_OK;
    });
    if (rc) return rc;
  } else if (small) {
    rc = timed_launch(ctx, "small_kernel", [&]() -> int {
      bool unsupported = false;
      hipError_t he = launch_small(a, src, grid, ctx->stream, &unsupported);
      if (unsupported) return fail(ctx, MIMO_E_UNSUPPORTED, "no small-shape kernel for K=%d, Dz=%d", K, D);
      HIP_TRY(ctx, he);
      return MIMO_OK;
    });
    if (rc) return rc;
  } else if (fused_covers(a.K16, ncb_total, src)) {
    rc = timed_launch(ctx, "fused_kernel", [&]() -> int {
      bool unsupported = false;
      hipError_t he = launch_fused(a, src, grid, ctx->stream, &unsupported);
      if (unsupported) return fail(ctx, MIMO_E_UNSUPPORTED, "no fused kernel for K=%d, Dz=%d", K, D);
      HIP_TRY(ctx, he);
      return MIMO_OK;
    });
    if (rc) return rc;
  } else {
    // two-stage path: chunked E-step writes responsibilities / labels, then the statistics kernel
    // runs once per group of <= kMaxNCB feature column blocks, all into the same partial block.
    KernelArgs st = a;
    int stats_src = src;
    if (src == kSrcEstep) {
      KernelArgs e = a;
      e.RS = 16 * kChunkNCB + 1;
      e.split = (a.split || a.resp || a.logp || a.lse) ? 1 : 0;   // include/mimo_hip.h: scalars[1..2] come with the split or any kept table
      if (!e.gibbs && e.do_stats && !e.resp) {       // statistics need the table: keep it internally
        const size_t kn = (size_t)K * (size_t)(ctx->N > 0 ? ctx->N : 1);
        if ((rc = ensure_dev(ctx, &ctx->resp, &ctx->resp_cap, kn))) return rc;
        e.resp = ctx->resp; ctx->resp_K = K; ctx->resp_valid = true;
      }
      if (chunked_lds_bytes(e) > 160 * 1024)
        return fail(ctx, MIMO_E_UNSUPPORTED, "K=%d, Dz=%d needs more LDS than one CU has", K, D);
      if (wide_estep_covers(a.K16, D, a.F16, e.gibbs)) {          // pipelined softmax / label pass (mimo_wide.hip)
        rc = timed_launch(ctx, "wide_estep_kernel", [&]() -> int {
          HIP_TRY(ctx, launch_wide_estep(e, grid, ctx->stream));
          return MIMO_OK;
        });
      } else {
        rc = timed_launch(ctx, "estep_chunked_kernel", [&]() -> int {
          HIP_TRY(ctx, launch_estep_chunked(e, grid, ctx->stream));
          return MIMO_OK;
        });
      }
      if (rc) return rc;
      st.resp = e.resp; st.labels = e.labels; st.write_scalars = 0;
      stats_src = e.gibbs ? kSrcLabels : kSrcWeights;
      if (ctx->n_bad > 0 && !e.gibbs && a.do_stats) {     // rows with NaN: their weights are dropped from the statistics
        const size_t kn = (size_t)K * (size_t)ctx->N;
        if ((rc = ensure_dev(ctx, &ctx->table_tmp, &ctx->table_tmp_cap, kn))) return rc;
        HIP_TRY(ctx, launch_mask_table(e.resp, ctx->row_mask, ctx->table_tmp, K, ctx->N, ctx->stream));
        st.resp = ctx->table_tmp;
      }
    }
    if (a.do_stats && stats_src == kSrcLabels && label_stats_covers(K, D, ctx->structure)) {
      // label-indexed statistics of the labels just drawn: the HBM-bound pass instead of one-hot products per column group
      KernelArgs g = st;
      g.gibbs = 0; g.do_stats = 1; g.logp = nullptr; g.lse = nullptr;
      if ((rc = prepare_label_presort(ctx, g))) return rc;
      rc = timed_launch(ctx, "label_stats_kernel", [&]() -> int {
        HIP_TRY(ctx, launch_label_stats(g, ctx->structure, grid, ctx->stream));
        return MIMO_OK;
      });
      if (rc) return rc;
    } else if (a.do_stats) {
      const bool wide = stats_src == kSrcWeights && wide_stats_covers(a.K16, D);     // 8-wave statistics kernel (mimo_wide.hip)
      const int gmax = wide ? wide_stats_group_ncb(a.K16, ncb_total) : stats_group_ncb(a.K16);
      for (int cb0 = 0; cb0 < ncb_total; cb0 += gmax) {
        KernelArgs g = st;
        const int ncb = ncb_total - cb0 < gmax ? ncb_total - cb0 : gmax;
        g.cb0 = cb0; g.F16 = 16 * ncb; g.RS = g.F16 + 1; g.F16_total = a.F16;
        g.gibbs = 0; g.do_stats = 1; g.logp = nullptr; g.lse = nullptr;
        if (cb0 > 0) g.write_scalars = 0;
        if (wide) {
          rc = timed_launch(ctx, "wide_stats_kernel", [&]() -> int {
            HIP_TRY(ctx, launch_wide_stats(g, grid, ctx->stream));
            return MIMO_OK;
          });
        } else {
          rc = timed_launch(ctx, src == kSrcEstep ? "fused_kernel(statistics of a column group)" : "fused_kernel", [&]() -> int {
            bool unsupported = false;
            HIP_TRY(ctx, launch_fused(g, stats_src, grid, ctx->stream, &unsupported));
            if (unsupported) return fail(ctx, MIMO_E_UNSUPPORTED, "no statistics kernel for K=%d, Dz=%d", K, D);
            return MIMO_OK;
          });
        }
        if (rc) return rc;
      }
    }
  }
  if (ctx->prof) ctx->prof_n += 1;
  const bool async = (flags & MIMO_F_ASYNC) != 0;
  const bool want_stats = a.do_stats && (S || async);
  const bool device_out = (flags & MIMO_F_DEVICE_OUT) != 0;
  if (!want_stats && !scalars && !async) return MIMO_OK;

  const size_t slen = (size_t)K * (1 + D + (size_t)D * D);
  // the small-shape kernel always accumulates the full feature map: under a structure hint the entries outside
  // the structure are masked to the zeros the hint promises
  const uint8_t* feat = small ? ctx->feat_full_d : ctx->feat_d;
  const int F = small ? feat_count(D) : ctx->F, mask = small ? ctx->structure : 0;
  if (device_out) {
    HIP_TRY(ctx, launch_reduce_unpack(ctx->partials, grid, (int64_t)pstride, feat, K, D, F, a.F16, want_stats ? S : nullptr, scalars,
                                      ctx->stream, mask));
    if (ctx->comm) {     // sharded through this library: sum over the ranks where the caller wants the block
      char msg[256];
      if (want_stats && (rc = mimo_comm::allreduce_sum_f64(ctx->comm, S, slen, ctx->stream, msg, sizeof msg))) return fail(ctx, rc, "%s", msg);
      if (scalars && (rc = mimo_comm::allreduce_sum_f64(ctx->comm, scalars, 3, ctx->stream, msg, sizeof msg))) return fail(ctx, rc, "%s", msg);
    }
    return MIMO_OK;
  }
  if ((rc = ensure_dev(ctx, &ctx->S_d, &ctx->S_cap, slen + 4))) return rc;
  if ((rc = ensure_pinned(ctx, &ctx->S_h, &ctx->S_hcap, slen + 4))) return rc;
  // Without a communicator and on the full feature map the reduction writes the packed block straight into the pinned host
  // buffer (device-visible, hipHostMalloc): no device copy of the block, no D2H transfer behind the kernel.
  static const bool direct_on = [] { const char* e = getenv("MIMO_DIRECT_OUT"); return !e || atoi(e) != 0; }();   // tuning knob
  const bool direct = direct_on && !ctx->comm && F == feat_count(D);
  double* dst = direct ? ctx->S_h : ctx->S_d;
  if (ctx->comm && !want_stats) HIP_TRY(ctx, hipMemsetAsync(ctx->S_d, 0, slen * sizeof(double), ctx->stream));
  HIP_TRY(ctx, launch_reduce_unpack(ctx->partials, grid, (int64_t)pstride, feat, K, D, F, a.F16, want_stats ? dst : nullptr,
                                    dst + slen, ctx->stream, mask));
  if (ctx->comm) {       // ONE all-reduce(sum, f64) of [K (1 + Dz + Dz^2) + 3] per pass, behind the kernels on the same stream
    char msg[256];
    if ((rc = mimo_comm::allreduce_sum_f64(ctx->comm, ctx->S_d, slen + 3, ctx->stream, msg, sizeof msg))) return fail(ctx, rc, "%s", msg);
  }
  if (!direct) HIP_TRY(ctx, hipMemcpyAsync(ctx->S_h, ctx->S_d, (slen + 4) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (flags & MIMO_F_ASYNC) {
    ctx->pending_async = true; ctx->pending_slen = slen; ctx->pending_stats = want_stats;
    return MIMO_OK;
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (want_stats) memcpy(S, ctx->S_h, slen * sizeof(double));
  if (scalars) memcpy(scalars, ctx->S_h + slen, 3 * sizeof(double));
  return MIMO_OK;
}

// Every pass goes through here.  Data without NaN rows: straight to run_pass.  With NaN rows (zeroed in the library's
// copy, ctx->row_mask = 0 there): the log-densities, tables, labels and ELBO scalars are those of the zeroed rows
// (= the reference's normaliser-only log-density), the statistics leave those rows out —
//   softmax pass : the mask becomes the per-row weight vector of the statistics (times the caller's weights, if any);
//   label pass   : labels first (no statistics), then the statistics of the labels with the NaN rows set to -1; the
//                  labels drawn ON the NaN rows are counted per component for the gating update (mimo_nan_info);
//   statistics of a caller's table / labels: masked copies.
static int run_fused(mimo_ctx* ctx, KernelArgs& a, int src, int flags, double* S, double* scalars) {
  if (ctx->n_bad <= 0) return run_pass(ctx, a, src, flags, S, scalars);
  int rc;
  const int64_t N = ctx->N;
  ctx->bad_counts_K = 0;
  if (src == kSrcEstep && !a.gibbs) {
    if (!a.u) {
      a.u = ctx->row_mask;
    } else {       // caller's row weights x mask (one column of a "table")
      if ((rc = ensure_dev(ctx, &ctx->table_tmp, &ctx->table_tmp_cap, (size_t)N))) return rc;
      HIP_TRY(ctx, launch_mask_table(a.u, ctx->row_mask, ctx->table_tmp, 1, N, ctx->stream));
      a.u = ctx->table_tmp;
    }
    return run_pass(ctx, a, src, flags, S, scalars);
  }
  if ((src == kSrcEstep && a.gibbs) || src == kSrcLabels) {
    const bool want = a.do_stats != 0;
    if (src == kSrcEstep) {
      if (flags & MIMO_F_ASYNC) return fail(ctx, MIMO_E_UNSUPPORTED, "asynchronous label pass on data with NaN rows");
      a.do_stats = 0;
      if ((rc = run_pass(ctx, a, kSrcEstep, flags & ~(MIMO_F_DEVICE_OUT), nullptr, nullptr))) return rc;
      ctx->rowwave_call = false;
      ctx->narrow_call = 0;
      ctx->mid_labels_call = false;
    }
    if ((rc = ensure_dev(ctx, &ctx->labels_tmp, &ctx->labels_tmp_cap, (size_t)N))) return rc;
    HIP_TRY(ctx, hipMemsetAsync(ctx->cnt_d + 1, 0, 256 * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, launch_mask_labels(a.labels, ctx->row_mask, ctx->labels_tmp, N, a.K, ctx->cnt_d + 1, ctx->stream));
    ctx->bad_counts_K = a.K;
    if (!want) return MIMO_OK;
    KernelArgs b = a;
    b.labels = ctx->labels_tmp; b.gibbs = 0; b.do_stats = 1; b.u = nullptr; b.logp = nullptr; b.lse = nullptr; b.resp = nullptr;
    return run_pass(ctx, b, kSrcLabels, flags, S, scalars);
  }
  // kSrcWeights: statistics of a (K, N) table
  const size_t kn = (size_t)a.K * (size_t)N;
  if ((rc = ensure_dev(ctx, &ctx->table_tmp, &ctx->table_tmp_cap, kn))) return rc;
  HIP_TRY(ctx, launch_mask_table(a.resp, ctx->row_mask, ctx->table_tmp, a.K, N, ctx->stream));
  a.resp = ctx->table_tmp;
  return run_pass(ctx, a, src, flags, S, scalars);
}

// ------------------------------------------------------------------------------------------
extern "C" {

const char* mimo_version(void) { return "mimo_hip 0.1 (gfx950, f64 MFMA feature-GEMM)"; }

int mimo_create(mimo_ctx** out, int device) {
  return guarded(nullptr, [&]() -> int {
  if (!out) return fail(nullptr, MIMO_E_INVALID, "mimo_create: out is NULL");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(nullptr, MIMO_E_HIP, "mimo_create: no HIP device available (%s)", hipGetErrorString(e));
  if (device < 0 || device >= ndev)
    return fail(nullptr, MIMO_E_INVALID, "mimo_create: device %d out of range [0,%d)", device, ndev);
  mimo_ctx* ctx = new (std::nothrow) mimo_ctx();
  if (!ctx) return fail(nullptr, MIMO_E_INVALID, "mimo_create: out of host memory");
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete ctx; return fail(nullptr, MIMO_E_HIP, "hipSetDevice failed"); }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cu = ctx->hw_num_cu = prop.multiProcessorCount;
  if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return fail(nullptr, MIMO_E_HIP, "hipStreamCreate failed");
  }
  ctx->stream = ctx->own_stream;
  if (hipMalloc(reinterpret_cast<void**>(&ctx->ls_aux), label_stats_aux_words() * sizeof(uint32_t)) != hipSuccess) {
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return fail(nullptr, MIMO_E_HIP, "hipMalloc failed");
  }
  *out = ctx;
  return MIMO_OK;
  });
}

int mimo_destroy(mimo_ctx* ctx) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MIMO_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  drain_profile(ctx);
  if (ctx->comm) { (void)mimo_comm::destroy(ctx->comm); ctx->comm = nullptr; }
  void* bufs[] = {ctx->sort_list, ctx->sort_start, ctx->ls_aux, ctx->Z_owned, ctx->feat_d, ctx->feat_full_d, ctx->row_mask, ctx->cnt_d, ctx->labels_tmp, ctx->table_tmp, ctx->theta_d, ctx->partials, ctx->reduced, ctx->S_d, ctx->resp,
                  ctx->logp, ctx->lse, ctx->labels, ctx->u_d, ctx->win, ctx->lin};
  for (void* p : bufs) if (p) (void)hipFree(p);
  if (ctx->theta_h) (void)hipHostFree(ctx->theta_h);
  if (ctx->S_h) (void)hipHostFree(ctx->S_h);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
  return MIMO_OK;
  });
}

const char* mimo_last_error(const mimo_ctx* ctx) { return ctx ? ctx->err : g_err; }

int mimo_set_stream(mimo_ctx* ctx, void* hip_stream) {
  return guarded(ctx, [&]() -> int {
  int rc = bind(ctx); if (rc) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
  return MIMO_OK;
  });
}

static int set_data(mimo_ctx* ctx, int64_t N, int Dz) {
  if (N < 0) return fail(ctx, MIMO_E_INVALID, "N must be >= 0");
  if (Dz < 1 || Dz > kMaxD)
    return fail(ctx, MIMO_E_UNSUPPORTED, "Dz = %d outside [1, %d]", Dz, kMaxD);
  ctx->N = N; ctx->D = Dz;
  ctx->resp_valid = ctx->logp_valid = ctx->lse_valid = ctx->labels_valid = false;
  ctx->weights_resident = false;
  return prepare_features(ctx, Dz);
}

// Find the rows that hold a NaN.  `owned`: Z is the library's copy — such rows are zeroed in place and the mask written;
// otherwise (borrowed device buffer) they are only counted, and if there are any the data is copied first.
static int scan_nan_rows(mimo_ctx* ctx, bool checksum) {
  ctx->n_bad = 0; ctx->bad_counts_K = 0;
  ctx->data_sum_valid = false;
  if (ctx->N <= 0) {
    if (checksum) { ctx->data_sum[0] = ctx->data_sum[1] = 0; ctx->data_sum_valid = true; }
    return MIMO_OK;
  }
  int rc;
  if (!ctx->cnt_d) HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->cnt_d), 259 * sizeof(unsigned long long)));
  // flat scan first (one coalesced read of Z): data without a NaN — the usual case — is done after it; the same read
  // yields the content checksum of an upload (before any row is zeroed)
  HIP_TRY(ctx, hipMemsetAsync(ctx->cnt_d, 0, sizeof(unsigned long long), ctx->stream));
  if (checksum) HIP_TRY(ctx, hipMemsetAsync(ctx->cnt_d + 257, 0, 2 * sizeof(unsigned long long), ctx->stream));
  HIP_TRY(ctx, launch_nan_any(ctx->Z, ctx->N * ctx->D, reinterpret_cast<unsigned int*>(ctx->cnt_d), ctx->hw_num_cu, ctx->stream,
                              checksum ? ctx->cnt_d + 257 : nullptr));
  unsigned long long nb = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&nb, ctx->cnt_d, sizeof nb, hipMemcpyDeviceToHost, ctx->stream));
  if (checksum) HIP_TRY(ctx, hipMemcpyAsync(ctx->data_sum, ctx->cnt_d + 257, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->data_sum_valid = checksum;
  if (nb == 0) return MIMO_OK;
  HIP_TRY(ctx, hipMemsetAsync(ctx->cnt_d, 0, sizeof(unsigned long long), ctx->stream));
  HIP_TRY(ctx, launch_nan_scan(const_cast<double*>(ctx->Z), ctx->N, ctx->D, nullptr, ctx->cnt_d, false, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(&nb, ctx->cnt_d, sizeof nb, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (nb == 0) return MIMO_OK;
  if (ctx->Z != ctx->Z_owned) {        // borrowed buffer: never written — work on a copy
    const size_t bytes = (size_t)ctx->N * ctx->D * sizeof(double);
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->Z_owned), bytes));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->Z_owned, ctx->Z, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    ctx->Z = ctx->Z_owned;
  }
  if ((rc = ensure_dev(ctx, &ctx->row_mask, &ctx->mask_cap, (size_t)ctx->N))) return rc;
  HIP_TRY(ctx, hipMemsetAsync(ctx->cnt_d, 0, sizeof(unsigned long long), ctx->stream));
  HIP_TRY(ctx, launch_nan_scan(ctx->Z_owned, ctx->N, ctx->D, ctx->row_mask, ctx->cnt_d, true, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->n_bad = (int64_t)nb;
  return MIMO_OK;
}

int mimo_upload(mimo_ctx* ctx, const double* Z_host, int64_t N, int Dz) {
  return guarded(ctx, [&]() -> int {
  int rc = bind(ctx); if (rc) return rc;
  if (!Z_host && N > 0) return fail(ctx, MIMO_E_INVALID, "mimo_upload: Z is NULL");
  if ((rc = set_data(ctx, N, Dz))) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->Z_owned) { HIP_TRY(ctx, hipFree(ctx->Z_owned)); ctx->Z_owned = nullptr; }
  const size_t bytes = (size_t)(N > 0 ? N : 1) * Dz * sizeof(double);
  HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->Z_owned), bytes));
  if (N > 0) HIP_TRY(ctx, hipMemcpy(ctx->Z_owned, Z_host, (size_t)N * Dz * sizeof(double), hipMemcpyHostToDevice));
  ctx->Z = ctx->Z_owned;
  return scan_nan_rows(ctx, true);
  });
}

int mimo_data_checksum(mimo_ctx* ctx, uint64_t out[2]) {
  return guarded(ctx, [&]() -> int {
  if (!ctx || !out) return fail(ctx, MIMO_E_INVALID, "mimo_data_checksum: null argument");
  if (!ctx->Z || !ctx->data_sum_valid) return fail(ctx, MIMO_E_STATE, "mimo_data_checksum: no uploaded rows (attached device rows are not summed)");
  out[0] = ctx->data_sum[0]; out[1] = ctx->data_sum[1];
  return MIMO_OK;
  });
}

int mimo_attach(mimo_ctx* ctx, const double* Z_dev, int64_t N, int Dz) {
  return guarded(ctx, [&]() -> int {
  int rc = bind(ctx); if (rc) return rc;
  if (!Z_dev) return fail(ctx, MIMO_E_INVALID, "mimo_attach: Z is NULL");
  if ((rc = set_data(ctx, N, Dz))) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->Z_owned) { HIP_TRY(ctx, hipFree(ctx->Z_owned)); ctx->Z_owned = nullptr; }
  ctx->Z = Z_dev;
  return scan_nan_rows(ctx, false);
  });
}

int mimo_set_structure(mimo_ctx* ctx, int structure) {
  return guarded(ctx, [&]() -> int {
  int rc = bind(ctx); if (rc) return rc;
  if (structure != MIMO_STRUCT_FULL && structure != MIMO_STRUCT_DIAG && structure != MIMO_STRUCT_LINEAR)
    return fail(ctx, MIMO_E_INVALID, "mimo_set_structure: unknown structure %d", structure);
  if (ctx->pending_async) return fail(ctx, MIMO_E_STATE, "an asynchronous call is pending: call mimo_wait first");
  if (ctx->structure == structure) return MIMO_OK;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));    // the feature table may be in use
  ctx->structure = structure;
  return ctx->D > 0 ? prepare_features(ctx, ctx->D) : MIMO_OK;
  });
}

int mimo_set_row_offset(mimo_ctx* ctx, int64_t row0) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return fail(nullptr, MIMO_E_INVALID, "null context");
  ctx->row0 = row0;
  return MIMO_OK;
  });
}

static int keep_tables(mimo_ctx* ctx, int K, int flags, KernelArgs* a) {
  int rc;
  const size_t kn = (size_t)K * (size_t)(ctx->N > 0 ? ctx->N : 1);
  if (flags & MIMO_F_KEEP_RESP) {
    if ((rc = ensure_dev(ctx, &ctx->resp, &ctx->resp_cap, kn))) return rc;
    a->resp = ctx->resp; ctx->resp_K = K; ctx->resp_valid = true;
  }
  if (flags & MIMO_F_KEEP_LOGP) {
    if ((rc = ensure_dev(ctx, &ctx->logp, &ctx->logp_cap, kn))) return rc;
    a->logp = ctx->logp; ctx->logp_K = K; ctx->logp_valid = true;
  }
  if (flags & MIMO_F_KEEP_LSE) {
    if ((rc = ensure_dev(ctx, &ctx->lse, &ctx->lse_cap, (size_t)(ctx->N > 0 ? ctx->N : 1)))) return rc;
    a->lse = ctx->lse; ctx->lse_valid = true;
  }
  return MIMO_OK;
}

// Which bound-only requests run as the plain pass (mimo_estep).  MIMO_BOUND_PROMOTE = 0: none, 2: every shape of the narrow / mid kernels.
static bool bound_promote(const mimo_ctx* ctx, int K) {
  static const int mode = [] { const char* e = getenv("MIMO_BOUND_PROMOTE"); return e ? atoi(e) : 1; }();
  if (mode == 0) return false;
  const bool md = use_mid(ctx, K, true);
  const bool nv = !md && use_narrow(ctx, K, false, true, true) != 0;
  if (mode == 2) return md || nv;
  return (md && K <= g_bound_promote_mid_k) || nv;
}

int mimo_estep(mimo_ctx* ctx, const double* c, const double* b, const double* W, int K,
               int flags, double* S, double* scalars) {
  return guarded(ctx, [&]() -> int {
  int rc = bind(ctx); if (rc) return rc;
  if ((rc = check_shapes(ctx, K))) return rc;
  if (!c || !b || !W) return fail(ctx, MIMO_E_INVALID, "mimo_estep: c, b, W must be non-NULL");
  const bool no_stats = (flags & MIMO_F_NO_STATS) != 0;
  if (!no_stats && !S && !(flags & MIMO_F_ASYNC))
    return fail(ctx, MIMO_E_INVALID, "mimo_estep: S is NULL without MIMO_F_NO_STATS / MIMO_F_ASYNC");
  KernelArgs a;
  fill_args(ctx, K, &a);
  const bool tables = (flags & (MIMO_F_KEEP_RESP | MIMO_F_KEEP_LOGP | MIMO_F_KEEP_LSE | MIMO_F_ENTROPY_SPLIT)) != 0;
  // A bound-only request (no statistics, no tables: the full-data pass of every SVI outer iteration, gmm.py:319-326 / ilr.py:270-277
  // of the reference) used to take the generic tile kernels whatever the shape; where the plain pass runs on the narrow or mid kernels
  // those pay for 16 x 16 padding the plain pass does not have, and the plain pass with its statistics left in the partial blocks is
  // the faster bound (bound_promote(): measured rule).
  const bool promote = no_stats && !tables && !(flags & MIMO_F_DEVICE_OUT) && bound_promote(ctx, K);
  a.do_stats = (no_stats && !promote) ? 0 : 1;
  a.split = (flags & MIMO_F_ENTROPY_SPLIT) ? 1 : 0;
  if ((rc = keep_tables(ctx, K, flags, &a))) return rc;
  // plain softmax + statistics pass at K <= 64, Dz <= 9: the row-owner kernel (Theta in the row-owner image)
  const bool plain = (!no_stats || promote) && !tables;
  const bool md = use_mid(ctx, K, plain);                     // mid shapes (K <= 32 over wide rows): mimo_mid.hip
  const bool nv = !md && use_narrow(ctx, K, false, plain, true) != 0;      // narrow shapes (Dz <= 4, 32 < K <= 128; few components over many features): mimo_narrow.hip
  const bool rv = !nv && !md && plain && ctx->n_bad == 0 && ctx->D <= 16 && !use_small(ctx, K) && vi_rowwave_covers(K, ctx->F16, a.ZS);
  if ((rc = nv ? upload_theta_narrow(ctx, c, b, W, K, 0) : md ? upload_theta_mid(ctx, c, b, W, K)
            : rv ? upload_theta_rowwave(ctx, c, b, W, K) : upload_theta(ctx, c, b, W, K, &a))) return rc;
  a.theta = ctx->theta_d;
  ctx->rowwave_vi_call = rv;
  ctx->narrow_call = nv ? 1 : 0;
  ctx->mid_call = md;
  rc = run_fused(ctx, a, kSrcEstep, flags, no_stats ? nullptr : S, scalars);
  ctx->rowwave_vi_call = false;
  ctx->narrow_call = 0;
  ctx->mid_call = false;
  return rc;
  });
}

int mimo_estep_weighted(mimo_ctx* ctx, const double* c, const double* b, const double* W, int K,
                        const double* row_weights, int flags, double* S, double* scalars) {
  return guarded(ctx, [&]() -> int {
  int rc = bind(ctx); if (rc) return rc;
  if ((rc = check_shapes(ctx, K))) return rc;
  if (!c || !b || !W || (!row_weights && !(flags & MIMO_F_WEIGHTS_RESIDENT)))
    return fail(ctx, MIMO_E_INVALID, "mimo_estep_weighted: c, b, W, row_weights must be non-NULL");
  if (flags & MIMO_F_NO_STATS) return fail(ctx, MIMO_E_INVALID, "mimo_estep_weighted: the weights only enter the statistics");
  if (!S && !(flags & MIMO_F_ASYNC)) return fail(ctx, MIMO_E_INVALID, "mimo_estep_weighted: S is NULL");
  KernelArgs a;
  fill_args(ctx, K, &a);
  // the narrow kernels take the weights on their normaliser (plain requests: statistics + scalars only)
  const bool plain_w = (flags & (MIMO_F_KEEP_RESP | MIMO_F_KEEP_LOGP | MIMO_F_KEEP_LSE | MIMO_F_ENTROPY_SPLIT)) == 0;
  const bool md = use_mid(ctx, K, plain_w);              // mid shapes (K <= 32 over wide rows)
  const bool nv = !md && use_narrow(ctx, K, false, plain_w, true) != 0;
  if (!nv && !md && !fused_covers(a.K16, a.F16 / 16, kSrcEstep))
    return fail(ctx, MIMO_E_UNSUPPORTED, "mimo_estep_weighted: K=%d, Dz=%d runs on the two-stage path, which takes "
                "its weights as a table (mimo_estep + mimo_weighted_stats)", K, ctx->D);
  a.split = (flags & MIMO_F_ENTROPY_SPLIT) ? 1 : 0;
  if ((rc = keep_tables(ctx, K, flags, &a))) return rc;
  if (flags & MIMO_F_WEIGHTS_RESIDENT) {
    if (!ctx->weights_resident) return fail(ctx, MIMO_E_STATE, "mimo_estep_weighted: no row weights are resident on the device");
    a.u = ctx->u_d;
  } else if (flags & MIMO_F_DEVICE_IN) {
    a.u = row_weights;
  } else {
    const size_t n1 = (size_t)(ctx->N > 0 ? ctx->N : 1);
    if ((rc = ensure_dev(ctx, &ctx->u_d, &ctx->u_cap, n1))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->u_d, row_weights, (size_t)ctx->N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // pageable host memory
    a.u = ctx->u_d;
    ctx->weights_resident = true;
  }
  if ((rc = nv ? upload_theta_narrow(ctx, c, b, W, K, 0) : md ? upload_theta_mid(ctx, c, b, W, K) : upload_theta(ctx, c, b, W, K, &a))) return rc;
  a.theta = ctx->theta_d;
  ctx->narrow_call = nv ? 1 : 0;
  ctx->mid_call = md;
  rc = run_fused(ctx, a, kSrcEstep, flags, S, scalars);
  ctx->narrow_call = 0;
  ctx->mid_call = false;
  return rc;
  });
}

int mimo_wait(mimo_ctx* ctx, double* S, double* scalars) {
  return guarded(ctx, [&]() -> int {
  int rc = bind(ctx); if (rc) return rc;
  if (!ctx->pending_async) return fail(ctx, MIMO_E_STATE, "mimo_wait: no asynchronous call is pending");
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->pending_async = false;
  if (S) {
    if (!ctx->pending_stats) return fail(ctx, MIMO_E_STATE, "mimo_wait: the pending call produced no statistics");
    memcpy(S, ctx->S_h, ctx->pending_slen * sizeof(double));
  }
  if (scalars) memcpy(scalars, ctx->S_h + ctx->pending_slen, 3 * sizeof(double));
  return MIMO_OK;
  });
}

int mimo_gibbs_labels(mimo_ctx* ctx, const double* c, const double* b, const double* W, int K,
                      uint64_t seed, uint64_t sweep, const double* u, int flags,
                      int32_t* labels_out, double* S) {
  return guarded(ctx, [&]() -> int {
  int rc = bind(ctx); if (rc) return rc;
  if ((rc = check_shapes(ctx, K))) return rc;
  if (!c || !b || !W) return fail(ctx, MIMO_E_INVALID, "mimo_gibbs_labels: c, b, W must be non-NULL");
  const bool no_stats = (flags & MIMO_F_NO_STATS) != 0 || !S;
  KernelArgs a;
  fill_args(ctx, K, &a);
  a.gibbs = 1;
  a.do_stats = no_stats ? 0 : 1;
  a.seed = seed; a.sweep = sweep;
  const size_t n1 = (size_t)(ctx->N > 0 ? ctx->N : 1);
  if ((rc = ensure_dev(ctx, &ctx->labels, &ctx->labels_cap, n1))) return rc;
  a.labels = ctx->labels; ctx->labels_valid = true;
  if ((rc = keep_tables(ctx, K, flags & (MIMO_F_KEEP_LOGP | MIMO_F_KEEP_LSE), &a))) return rc;
  if (u) {
    if (flags & MIMO_F_DEVICE_IN) {
      a.u = u;
    } else {
      if ((rc = ensure_dev(ctx, &ctx->u_d, &ctx->u_cap, n1))) return rc;
      HIP_TRY(ctx, hipMemcpyAsync(ctx->u_d, u, (size_t)ctx->N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // u is pageable host memory
      a.u = ctx->u_d;
      ctx->weights_resident = false;
    }
  }
  const bool wants_tables = (flags & (MIMO_F_KEEP_LOGP | MIMO_F_KEEP_LSE)) != 0;
  int nw = use_narrow(ctx, K, true, !wants_tables, !no_stats);
  const bool ml = !use_small(ctx, K) && use_mid_labels(ctx, K, wants_tables) && (!nw || mid_labels_before_narrow(ctx, K));
  if (ml) nw = 0;
  const bool rw = !nw && !ml && !use_small(ctx, K) && use_rowwave(ctx, K, wants_tables);
  if ((rc = nw ? upload_theta_narrow(ctx, c, b, W, K, nw - 1) : ml ? upload_theta_mid(ctx, c, b, W, K, true)
            : rw ? upload_theta_rowwave(ctx, c, b, W, K) : upload_theta(ctx, c, b, W, K, &a))) return rc;
  a.theta = ctx->theta_d;
  ctx->rowwave_call = rw;
  ctx->narrow_call = nw;
  ctx->mid_labels_call = ml;
  rc = run_fused(ctx, a, kSrcEstep, flags, no_stats ? nullptr : S, nullptr);
  ctx->rowwave_call = false;
  ctx->narrow_call = 0;
  ctx->mid_labels_call = false;
  if (rc) return rc;
  if (labels_out && !(flags & MIMO_F_DEVICE_OUT)) {
    HIP_TRY(ctx, hipMemcpyAsync(labels_out, ctx->labels, (size_t)ctx->N * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  return MIMO_OK;
  });
}

int mimo_weighted_stats(mimo_ctx* ctx, const double* resp, int K, int flags, double* S) {
  return guarded(ctx, [&]() -> int {
  int rc = bind(ctx); if (rc) return rc;
  if ((rc = check_shapes(ctx, K))) return rc;
  if (!S) return fail(ctx, MIMO_E_INVALID, "mimo_weighted_stats: S is NULL");
  KernelArgs a;
  fill_args(ctx, K, &a);
  if (!resp) {
    if (!ctx->resp_valid || ctx->resp_K != K)
      return fail(ctx, MIMO_E_STATE, "mimo_weighted_stats: resp is NULL and no (K=%d,N) table is resident", K);
    a.resp = ctx->resp;
  } else if (flags & MIMO_F_DEVICE_IN) {
    a.resp = const_cast<double*>(resp);
  } else {
    const size_t kn = (size_t)K * (size_t)(ctx->N > 0 ? ctx->N : 1);
    if ((rc = ensure_dev(ctx, &ctx->win, &ctx->win_cap, kn))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->win, resp, (size_t)K * ctx->N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    a.resp = ctx->win;
  }
  return run_fused(ctx, a, kSrcWeights, flags, S, nullptr);
  });
}

int mimo_label_stats(mimo_ctx* ctx, const int32_t* labels, int K, int flags, double* S) {
  return guarded(ctx, [&]() -> int {
  int rc = bind(ctx); if (rc) return rc;
  if ((rc = check_shapes(ctx, K))) return rc;
  if (!S) return fail(ctx, MIMO_E_INVALID, "mimo_label_stats: S is NULL");
  KernelArgs a;
  fill_args(ctx, K, &a);
  if (!labels) {
    if (!ctx->labels_valid) return fail(ctx, MIMO_E_STATE, "mimo_label_stats: labels is NULL and none are resident");
    a.labels = ctx->labels;
  } else if (flags & MIMO_F_DEVICE_IN) {
    a.labels = const_cast<int32_t*>(labels);
  } else {
    const size_t n1 = (size_t)(ctx->N > 0 ? ctx->N : 1);
    if ((rc = ensure_dev(ctx, &ctx->lin, &ctx->lin_cap, n1))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->lin, labels, (size_t)ctx->N * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    a.labels = ctx->lin;
  }
  return run_fused(ctx, a, kSrcLabels, flags, S, nullptr);
  });
}

int mimo_sample_from_log(mimo_ctx* ctx, const double* logp, int K, int64_t N, const double* u, uint64_t seed,
                         uint64_t sweep, int flags, int32_t* labels_out, double* lognorms_out) {
  return guarded(ctx, [&]() -> int {
  int rc = bind(ctx); if (rc) return rc;
  if (ctx->pending_async) return fail(ctx, MIMO_E_STATE, "an asynchronous call is pending: call mimo_wait first");
  if (K < 1 || N < 0 || !labels_out) return fail(ctx, MIMO_E_INVALID, "mimo_sample_from_log: bad arguments");
  const double* table = logp;
  const size_t kn = (size_t)K * (size_t)(N > 0 ? N : 1), n1 = (size_t)(N > 0 ? N : 1);
  if (!logp) {
    if (!ctx->logp_valid || ctx->logp_K != K || ctx->N != N)
      return fail(ctx, MIMO_E_STATE, "mimo_sample_from_log: logp is NULL and no (K=%d, N) log-density table is resident", K);
    table = ctx->logp;
  } else if (!(flags & MIMO_F_DEVICE_IN)) {
    if ((rc = ensure_dev(ctx, &ctx->win, &ctx->win_cap, kn))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->win, logp, (size_t)K * N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    table = ctx->win;
  }
  const double* ud = nullptr;
  if (u) {
    if (flags & MIMO_F_DEVICE_IN) ud = u;
    else {
      if ((rc = ensure_dev(ctx, &ctx->u_d, &ctx->u_cap, n1))) return rc;
      HIP_TRY(ctx, hipMemcpyAsync(ctx->u_d, u, (size_t)N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      ud = ctx->u_d;
      ctx->weights_resident = false;
    }
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));     // pageable host sources
  if ((rc = ensure_dev(ctx, &ctx->lin, &ctx->lin_cap, n1))) return rc;
  double* ln_d = nullptr;
  if (lognorms_out) {
    if ((rc = ensure_dev(ctx, &ctx->lse, &ctx->lse_cap, n1 > (size_t)(ctx->N > 0 ? ctx->N : 1) ? n1 : (size_t)(ctx->N > 0 ? ctx->N : 1)))) return rc;
    ctx->lse_valid = false;      // (the buffer is borrowed: whatever log-normaliser it held is gone)
    ln_d = ctx->lse;
  }
  HIP_TRY(ctx, launch_sample_table(table, K, N, ud, seed, sweep, ctx->row0, ctx->lin, ln_d, ctx->stream));
  if (N > 0) {
    HIP_TRY(ctx, hipMemcpyAsync(labels_out, ctx->lin, (size_t)N * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (lognorms_out) HIP_TRY(ctx, hipMemcpyAsync(lognorms_out, ln_d, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MIMO_OK;
  });
}

int mimo_random_resp_stats(mimo_ctx* ctx, int K, uint64_t seed, int flags, double* S) {
  return guarded(ctx, [&]() -> int {
  int rc = bind(ctx); if (rc) return rc;
  if ((rc = check_shapes(ctx, K))) return rc;
  if (!S) return fail(ctx, MIMO_E_INVALID, "mimo_random_resp_stats: S is NULL");
  const size_t kn = (size_t)K * (size_t)(ctx->N > 0 ? ctx->N : 1);
  if ((rc = ensure_dev(ctx, &ctx->resp, &ctx->resp_cap, kn))) return rc;
  ctx->resp_K = K; ctx->resp_valid = true;
  HIP_TRY(ctx, launch_random_resp(ctx->resp, K, ctx->N, seed, ctx->row0, ctx->stream));
  KernelArgs a;
  fill_args(ctx, K, &a);
  a.resp = ctx->resp;
  return run_fused(ctx, a, kSrcWeights, flags, S, nullptr);
  });
}

int mimo_table_entropy(mimo_ctx* ctx, const double* table, int64_t count, int flags, double* out) {
  return guarded(ctx, [&]() -> int {
  int rc = bind(ctx); if (rc) return rc;
  if (!out || count < 0) return fail(ctx, MIMO_E_INVALID, "mimo_table_entropy: bad arguments");
  const double* src = table;
  if (!table) {
    if (!ctx->resp_valid) return fail(ctx, MIMO_E_STATE, "mimo_table_entropy: table is NULL and no resp table is resident");
    src = ctx->resp;
    count = (int64_t)ctx->resp_K * ctx->N;
  } else if (!(flags & MIMO_F_DEVICE_IN)) {
    if ((rc = ensure_dev(ctx, &ctx->win, &ctx->win_cap, (size_t)(count > 0 ? count : 1)))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->win, table, (size_t)count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    src = ctx->win;
  }
  const int nblocks = 1024;
  if ((rc = ensure_dev(ctx, &ctx->partials, &ctx->partials_cap, (size_t)nblocks))) return rc;
  if ((rc = ensure_dev(ctx, &ctx->reduced, &ctx->reduced_cap, 4))) return rc;
  HIP_TRY(ctx, launch_table_entropy(src, count, ctx->partials, nblocks, ctx->reduced, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(out, ctx->reduced, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MIMO_OK;
  });
}

int mimo_predict(mimo_ctx* ctx, const double* c, const double* b, const double* W, int K,
                 const double* M, const double* Q, const double* Cc, int dy, int affine, int mode,
                 const double* y, const double* P, const double* ld,
                 double* mu, double* covar, double* nlpd) {
  return mimo_predict_flags(ctx, c, b, W, K, M, Q, Cc, dy, affine, mode, y, P, ld, mu, covar, nlpd, 0);
}

int mimo_predict_flags(mimo_ctx* ctx, const double* c, const double* b, const double* W, int K,
                       const double* M, const double* Q, const double* Cc, int dy, int affine, int mode,
                       const double* y, const double* P, const double* ld,
                       double* mu, double* covar, double* nlpd, int flags) {
  return guarded(ctx, [&]() -> int {
  const bool dev_in = (flags & MIMO_F_DEVICE_IN) != 0, dev_out = (flags & MIMO_F_DEVICE_OUT) != 0, diag = (flags & MIMO_F_DIAG_VAR) != 0;
  const size_t ncov = diag ? 2 * (size_t)dy : (size_t)dy * dy;        // doubles of the second output per row
  int rc = bind(ctx); if (rc) return rc;
  if (!ctx->Z) return fail(ctx, MIMO_E_NODATA, "mimo_predict: no data resident (call mimo_upload)");
  if (!c || !b || !W || !M || !Q || !Cc || !mu || !covar || K < 1 || (mode != 0 && mode != 1))
    return fail(ctx, MIMO_E_INVALID, "mimo_predict: bad arguments");
  const bool want_nlpd = nlpd != nullptr;
  if (want_nlpd && (!y || !P || !ld)) return fail(ctx, MIMO_E_INVALID, "mimo_predict: nlpd needs y, P and ld");
  const int dx = ctx->D, dc = dx + (affine ? 1 : 0);
  const int64_t N = ctx->N;
  const size_t ng = (size_t)K * (1 + dx + (size_t)dx * dx), nM = (size_t)K * dy * dc, nQ = (size_t)K * dc * dc,
               nC = (size_t)K * dy * dy;
  const size_t nparam = ng + nM + nQ + 2 * nC + K;
  const size_t nout = dev_out ? 0 : (size_t)N * (dy + ncov + 1), nin = (want_nlpd && !dev_in) ? (size_t)N * dy : 0;
  // parameters | outputs | y, all in the staged-weights workspace
  if ((rc = ensure_dev(ctx, &ctx->win, &ctx->win_cap, nparam + nout + nin + 1))) return rc;
  std::vector<double> h(nparam);
  double* q = h.data();
  for (int k = 0; k < K; ++k) {
    *q++ = c[k];
    memcpy(q, b + (size_t)k * dx, sizeof(double) * dx); q += dx;
    memcpy(q, W + (size_t)k * dx * dx, sizeof(double) * dx * dx); q += (size_t)dx * dx;
  }
  memcpy(q, M, sizeof(double) * nM); q += nM;
  memcpy(q, Q, sizeof(double) * nQ); q += nQ;
  memcpy(q, Cc, sizeof(double) * nC); q += nC;
  if (want_nlpd) { memcpy(q, P, sizeof(double) * nC); memcpy(q + nC, ld, sizeof(double) * K); }
  else memset(q, 0, sizeof(double) * (nC + K));
  double* d = ctx->win;
  HIP_TRY(ctx, hipMemcpyAsync(d, h.data(), nparam * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  PredictArgs a{};
  a.Z = ctx->Z; a.N = N; a.dx = dx; a.dc = dc; a.dy = dy; a.K = K; a.mode = mode; a.diag = diag ? 1 : 0;
  a.gate = d; a.M = d + ng; a.Q = a.M + nM; a.Cc = a.Q + nQ; a.P = a.Cc + nC; a.ld = a.P + nC;
  double* out = d + nparam;
  if (dev_out) {
    a.mu = mu; a.covar = covar; a.nlpd = want_nlpd ? nlpd : nullptr;
  } else {
    a.mu = out; a.covar = out + (size_t)N * dy; a.nlpd = want_nlpd ? a.covar + (size_t)N * ncov : nullptr;
  }
  if (want_nlpd) {
    if (dev_in) {
      a.y = y;
    } else {
      double* yd = out + nout;
      HIP_TRY(ctx, hipMemcpyAsync(yd, y, nin * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      a.y = yd;
    }
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // h (pageable) must be consumed before it goes away
  bool unsupported = false;
  rc = timed_launch(ctx, "predict_kernel", [&]() -> int {
    HIP_TRY(ctx, launch_predict(a, ctx->stream, &unsupported));
    return MIMO_OK;
  });
  if (rc) return rc;
  if (unsupported) return fail(ctx, MIMO_E_UNSUPPORTED, "mimo_predict: dy=%d (max %d) or dx=%d not supported", dy, kMaxPredictDy, dx);
  if (ctx->prof) ctx->prof_n += 1;
  if (dev_out) return MIMO_OK;
  if (N > 0) {
    HIP_TRY(ctx, hipMemcpyAsync(mu, a.mu, (size_t)N * dy * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(covar, a.covar, (size_t)N * ncov * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (want_nlpd) HIP_TRY(ctx, hipMemcpyAsync(nlpd, a.nlpd, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MIMO_OK;
  });
}

static int copy_out(mimo_ctx* ctx, void* dst, const void* src, size_t bytes, bool valid, const char* what) {
  int rc = bind(ctx); if (rc) return rc;
  if (!dst) return fail(ctx, MIMO_E_INVALID, "%s: destination is NULL", what);
  if (!valid || !src) return fail(ctx, MIMO_E_STATE, "%s: table was never produced (pass the MIMO_F_KEEP_* flag)", what);
  HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MIMO_OK;
}

int mimo_get_resp(mimo_ctx* ctx, double* out) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return fail(nullptr, MIMO_E_INVALID, "null context");
  return copy_out(ctx, out, ctx->resp, (size_t)ctx->resp_K * ctx->N * sizeof(double), ctx->resp_valid, "mimo_get_resp");
  });
}
int mimo_get_resp_columns(mimo_ctx* ctx, const int64_t* cols, int64_t ncols, double* out) {
  return guarded(ctx, [&]() -> int {
  int rc = bind(ctx); if (rc) return rc;
  if (ncols < 0 || (ncols > 0 && (!cols || !out))) return fail(ctx, MIMO_E_INVALID, "mimo_get_resp_columns: bad arguments");
  if (!ctx->resp_valid || !ctx->resp) return fail(ctx, MIMO_E_STATE, "mimo_get_resp_columns: no responsibility table is resident");
  if (ncols == 0) return MIMO_OK;
  for (int64_t j = 0; j < ncols; ++j)
    if (cols[j] < 0 || cols[j] >= ctx->N) return fail(ctx, MIMO_E_INVALID, "mimo_get_resp_columns: column %lld outside [0, N)", (long long)cols[j]);
  const size_t K = (size_t)ctx->resp_K, words = (size_t)ncols + K * (size_t)ncols;      // indices | gathered block, in the staged-labels / weights workspace
  if ((rc = ensure_dev(ctx, &ctx->table_tmp, &ctx->table_tmp_cap, words))) return rc;
  int64_t* cols_d = reinterpret_cast<int64_t*>(ctx->table_tmp);
  double* out_d = ctx->table_tmp + ncols;
  HIP_TRY(ctx, hipMemcpyAsync(cols_d, cols, (size_t)ncols * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, launch_gather_columns(ctx->resp, (int)K, ctx->N, cols_d, ncols, out_d, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(out, out_d, K * (size_t)ncols * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MIMO_OK;
  });
}
int mimo_get_logp(mimo_ctx* ctx, double* out) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return fail(nullptr, MIMO_E_INVALID, "null context");
  return copy_out(ctx, out, ctx->logp, (size_t)ctx->logp_K * ctx->N * sizeof(double), ctx->logp_valid, "mimo_get_logp");
  });
}
int mimo_get_lse(mimo_ctx* ctx, double* out) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return fail(nullptr, MIMO_E_INVALID, "null context");
  return copy_out(ctx, out, ctx->lse, (size_t)ctx->N * sizeof(double), ctx->lse_valid, "mimo_get_lse");
  });
}
int mimo_get_labels(mimo_ctx* ctx, int32_t* out) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return fail(nullptr, MIMO_E_INVALID, "null context");
  return copy_out(ctx, out, ctx->labels, (size_t)ctx->N * sizeof(int32_t), ctx->labels_valid, "mimo_get_labels");
  });
}

int mimo_nan_info(mimo_ctx* ctx, int64_t* n_bad, double* row_mask_out, int K, int64_t* label_counts) {
  return guarded(ctx, [&]() -> int {
    int rc = bind(ctx); if (rc) return rc;
    if (n_bad) *n_bad = ctx->n_bad;
    if (row_mask_out && ctx->N > 0) {
      if (ctx->n_bad > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(row_mask_out, ctx->row_mask, (size_t)ctx->N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      } else {
        for (int64_t i = 0; i < ctx->N; ++i) row_mask_out[i] = 1.0;
      }
    }
    if (label_counts) {
      if (K < 1 || K > 256) return fail(ctx, MIMO_E_INVALID, "mimo_nan_info: K = %d outside [1, 256]", K);
      for (int k = 0; k < K; ++k) label_counts[k] = 0;
      if (ctx->n_bad > 0) {
        if (ctx->bad_counts_K != K) return fail(ctx, MIMO_E_STATE, "mimo_nan_info: no label pass with K = %d has run on this data", K);
        unsigned long long h[256];
        HIP_TRY(ctx, hipMemcpyAsync(h, ctx->cnt_d + 1, (size_t)K * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (int k = 0; k < K; ++k) label_counts[k] = (int64_t)h[k];
      }
    }
    return MIMO_OK;
  });
}

int mimo_shader_clock_mhz(mimo_ctx* ctx, double* mhz) {
  return guarded(ctx, [&]() -> int {
    int rc = bind(ctx); if (rc) return rc;
    if (!mhz) return fail(ctx, MIMO_E_INVALID, "mimo_shader_clock_mhz: out is NULL");
    const int grid = ctx->num_cu * 4;
    if ((rc = ensure_dev(ctx, &ctx->partials, &ctx->partials_cap, (size_t)grid * 2))) return rc;
    unsigned long long* d = reinterpret_cast<unsigned long long*>(ctx->partials);
    HIP_TRY(ctx, launch_clock_probe(d, grid, 20000, ctx->stream));      // ~0.3 ms of v_fma_f64 on every SIMD
    std::vector<unsigned long long> h((size_t)grid * 2);
    HIP_TRY(ctx, hipMemcpyAsync(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<double> v;
    for (int g = 0; g < grid; ++g)
      if (h[2 * g + 1] > 0) v.push_back((double)h[2 * g] / (double)h[2 * g + 1] * 100.0);
    if (v.empty()) return fail(ctx, MIMO_E_HIP, "mimo_shader_clock_mhz: no samples");
    std::sort(v.begin(), v.end());
    *mhz = v[v.size() / 2];
    return MIMO_OK;
  });
}

int mimo_comm_unique_id(char* id128) {
  return guarded(nullptr, [&]() -> int {
    if (!id128) return fail(nullptr, MIMO_E_INVALID, "mimo_comm_unique_id: id is NULL");
    char msg[256];
    const int rc = mimo_comm::unique_id(id128, msg, sizeof msg);
    return rc ? fail(nullptr, rc, "%s", msg) : MIMO_OK;
  });
}

int mimo_comm_init(mimo_ctx* ctx, const char* id128, int rank, int world) {
  return guarded(ctx, [&]() -> int {
    int rc = bind(ctx); if (rc) return rc;
    if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(ctx, MIMO_E_INVALID, "mimo_comm_init: bad arguments");
    if (ctx->pending_async) return fail(ctx, MIMO_E_STATE, "an asynchronous call is pending: call mimo_wait first");
    if (ctx->comm) { (void)mimo_comm::destroy(ctx->comm); ctx->comm = nullptr; }
    char msg[256];
    void* c = nullptr;
    if ((rc = mimo_comm::init(&c, id128, rank, world, msg, sizeof msg))) return fail(ctx, rc, "%s", msg);
    ctx->comm = c; ctx->comm_world = world;
    return MIMO_OK;
  });
}

int mimo_comm_destroy(mimo_ctx* ctx) {
  return guarded(ctx, [&]() -> int {
    int rc = bind(ctx); if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->comm) { (void)mimo_comm::destroy(ctx->comm); ctx->comm = nullptr; ctx->comm_world = 1; }
    return MIMO_OK;
  });
}

int mimo_debug_fault(mimo_ctx* ctx, int kind) {
  return guarded(ctx, [&]() -> int {
    if (kind == 1) throw std::bad_alloc();
    if (kind == 2) throw std::runtime_error("mimo_debug_fault");
    if (kind == 3) throw 42;
    return kind == 0 ? MIMO_OK : fail(ctx, MIMO_E_INVALID, "mimo_debug_fault: unknown kind %d", kind);
  });
}

int mimo_tune(mimo_ctx* ctx, const char* key, int64_t value) {
  return guarded(ctx, [&]() -> int {
    if (!ctx || !key) return fail(ctx, MIMO_E_INVALID, "mimo_tune: null argument");
    if (ctx->pending_async) return fail(ctx, MIMO_E_STATE, "an asynchronous call is pending: call mimo_wait first");
    if (!strcmp(key, "num_cu")) {
      if (value < 0 || value > 4096) return fail(ctx, MIMO_E_INVALID, "mimo_tune: num_cu = %lld outside [0, 4096]", (long long)value);
      ctx->num_cu = value == 0 ? ctx->hw_num_cu : (int)value;
      return MIMO_OK;
    }
    if (!strcmp(key, "sorted_range")) {
      if (value < 0 || value > 80) return fail(ctx, MIMO_E_INVALID, "mimo_tune: sorted_range = %lld outside [0, 80]", (long long)value);
      set_sorted_range_cap((int)value);
      return MIMO_OK;
    }
    if (!strcmp(key, "narrow_big_vi")) {
      if (value < 0 || value > 256) return fail(ctx, MIMO_E_INVALID, "mimo_tune: narrow_big_vi = %lld outside [0, 256]", (long long)value);
      set_narrow_big_vi((int)value);
      return MIMO_OK;
    }
    if (!strcmp(key, "mid_labels_narrow_k")) {
      if (value < 0 || value > 64) return fail(ctx, MIMO_E_INVALID, "mimo_tune: mid_labels_narrow_k = %lld outside [0, 64]", (long long)value);
      g_mid_labels_narrow_k = (int)value;
      return MIMO_OK;
    }
    if (!strcmp(key, "mid_labels_min_d")) {
      if (value < 0 || value > 64) return fail(ctx, MIMO_E_INVALID, "mimo_tune: mid_labels_min_d = %lld outside [0, 64]", (long long)value);
      g_mid_labels_min_d = (int)value;
      return MIMO_OK;
    }
    if (!strcmp(key, "mid_narrow_k")) {
      if (value < 0 || value > 64) return fail(ctx, MIMO_E_INVALID, "mimo_tune: mid_narrow_k = %lld outside [0, 64]", (long long)value);
      g_mid_narrow_k = (int)value;
      return MIMO_OK;
    }
    if (!strcmp(key, "mid_min_d")) {
      if (value < 0 || value > 64) return fail(ctx, MIMO_E_INVALID, "mimo_tune: mid_min_d = %lld outside [0, 64]", (long long)value);
      g_mid_min_d = (int)value;
      return MIMO_OK;
    }
    return fail(ctx, MIMO_E_INVALID, "mimo_tune: unknown key '%s'", key);
  });
}

double mimo_philox_uniform(uint64_t seed, uint64_t row, uint64_t sweep) {
  return philox_uniform_host(seed, row, sweep);
}

int mimo_profile(mimo_ctx* ctx, int enable) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return fail(nullptr, MIMO_E_INVALID, "null context");
  ctx->prof = enable != 0;
  return MIMO_OK;
  });
}

int mimo_profile_read(mimo_ctx* ctx, double* kernel_ms, int64_t* launches, int reset) {
  return guarded(ctx, [&]() -> int {
  int rc = bind(ctx); if (rc) return rc;
  drain_profile(ctx);
  if (kernel_ms) *kernel_ms = ctx->prof_ms;
  if (launches) *launches = ctx->prof_n;
  if (reset) {
    ctx->prof_ms = 0.0; ctx->prof_n = 0;
    for (int i = 0; i < mimo_ctx::kProfNames; ++i) { ctx->prof_name_ms[i] = 0.0; ctx->prof_name_n[i] = 0; }
  }
  return MIMO_OK;
  });
}

int mimo_profile_kernels(mimo_ctx* ctx, char* buf, int len) {
  return guarded(ctx, [&]() -> int {
  int rc = bind(ctx); if (rc) return rc;
  if (!buf || len < 1) return fail(ctx, MIMO_E_INVALID, "mimo_profile_kernels: no buffer");
  drain_profile(ctx);
  int off = 0;
  buf[0] = 0;
  for (int i = 0; i < mimo_ctx::kProfNames && ctx->prof_name[i]; ++i) {
    if (!ctx->prof_name_n[i]) continue;
    const int w = snprintf(buf + off, (size_t)(len - off), "%s\t%.6f\t%lld\n", ctx->prof_name[i], ctx->prof_name_ms[i],
                           (long long)ctx->prof_name_n[i]);
    if (w < 0 || w >= len - off) break;
    off += w;
  }
  return MIMO_OK;
  });
}

// The routing decision of mimo_plan without anything that needs a device: kind, launches, passes (out8[6] — the grid — stays 0),
// and a one-line description of the kernels (desc, may be null).
static void plan_route(const mimo_ctx* ctx, const KernelArgs& a, int K, int gibbs, int64_t* out8, char* desc, size_t dlen) {
  const int ncb = a.F16 / 16, D = ctx->D;
  char lst[160] = "";           // the label-statistics stage of a label pass
  auto label_stage = [&]() {
    const int ll = label_stats_launches(K, D, ctx->structure);
    if (label_stats_uses_slots(K, D, a.N)) snprintf(lst, sizeof lst, "label_slots_kernel + label_stats_slots_kernel");
    else if (ctx->structure == 0 && D >= 10 && label_stats_sorted(K, D, ctx->structure)) snprintf(lst, sizeof lst, "label_tile_sort_kernel + label_stats_sorted_kernel");
    else if (ctx->structure != 0 || D <= 9) snprintf(lst, sizeof lst, "label_stats_kernel");
    else if (D <= 16 && ll * 128 >= K && (K <= 64 || ll == (K + 127) / 128)) snprintf(lst, sizeof lst, "label_stats_wide_kernel x %d", ll);
    else snprintf(lst, sizeof lst, "label_stats_xwide_kernel x %d", ll);
    return ll;
  };
  memset(out8, 0, 8 * sizeof(int64_t));
  out8[4] = 1;                           // passes over Z
  out8[5] = gibbs ? 1 : 0;               // passes over the labels
  if (!gibbs && use_mid(ctx, K, true)) {
    out8[0] = MIMO_PLAN_MID; out8[1] = 1;
    if (desc) snprintf(desc, dlen, "mid_kernel<Dz=%d, row blocks %d, %d waves>", D, (K + 15) / 16, mid_rows_per_step(K, D) / 16);
  } else if (gibbs && use_mid_labels(ctx, K, false) && !use_small(ctx, K) && (mid_labels_before_narrow(ctx, K) || !use_narrow(ctx, K, true, true, true))) {
    const int ll = label_stage();
    out8[0] = MIMO_PLAN_MID; out8[1] = 1 + ll; out8[4] = 1 + ll; out8[5] = 1 + ll;
    if (desc) snprintf(desc, dlen, "mid_kernel<Dz=%d, row blocks %d, label draw> + %s", D, (K + 15) / 16, lst);
  } else if (const int nm = use_narrow(ctx, K, gibbs != 0, true, true)) {
    out8[0] = MIMO_PLAN_NARROW; out8[1] = nm == 2 ? 2 : 1;
    if (nm == 2) { out8[4] = 2; out8[5] = 2; }       // label kernel + label-statistics kernel (nm == 3: one kernel, labels written once)
    if (nm == 2) label_stage();
    if (desc) snprintf(desc, dlen, "narrow_kernel<%d slots, %d steps%s, %s>%s%s", narrow_v(K), narrow_steps(K, ctx->F, D, nm - 1),
                       narrow_dt(K, ctx->F, D, nm - 1) ? ", grouped" : "", nm == 1 ? "softmax + statistics" : nm == 2 ? "label draw" : "label draw + statistics",
                       nm == 2 ? " + " : "", nm == 2 ? lst : "");
  } else if (use_small(ctx, K)) {
    out8[0] = MIMO_PLAN_SMALL; out8[1] = 1;
    if (desc) snprintf(desc, dlen, "small_kernel<%d components per lane, %d lanes per row>", small_kl(D, K), small_g(D, K));
  } else if (!gibbs && ctx->n_bad == 0 && D <= 16 && vi_rowwave_covers(K, ctx->F16, a.ZS)) {
    out8[0] = MIMO_PLAN_ROWWAVE_VI; out8[1] = 1;
    if (desc) snprintf(desc, dlen, "vi_rowwave_kernel<%d row blocks>", K <= 32 ? 2 : 4);
  } else if (gibbs && use_rowwave(ctx, K, false)) {
    const int ll = label_stage();     // (> 1: the sliced statistics of the large shapes)
    out8[0] = MIMO_PLAN_ROWWAVE; out8[1] = 1 + ll;
    out8[4] = 1 + ll;                    // Z: label kernel + every statistics launch
    out8[5] = 1 + ll;                    // labels written once, read once per statistics launch
    if (desc) snprintf(desc, dlen, "%s<%d row blocks> + %s", gibbs_rowwave_counts_labels(K, ctx->F16, a.ZS) ? "gibbs_rowwave_kernel" : "gibbs_stream_kernel",
                       rowwave_kb_shape(K, ctx->F16, a.ZS), lst);
  } else if (fused_covers(a.K16, ncb, kSrcEstep)) {
    out8[0] = MIMO_PLAN_FUSED; out8[1] = 1;
    if (desc) snprintf(desc, dlen, "fused_kernel<%d column blocks, %d row block%s per wave%s>", ncb, a.K16 <= 4 ? 1 : a.K16 <= 8 ? 2 : a.K16 <= 12 ? 3 : 4,
                       a.K16 <= 4 ? "" : "s", K <= 32 && D >= 7 ? ", work split over the waves" : "");
  } else {
    const bool wide = !gibbs && wide_stats_covers(a.K16, D);       // as run_pass
    const int gmax = wide ? wide_stats_group_ncb(a.K16, ncb) : stats_group_ncb(a.K16), groups = (ncb + gmax - 1) / gmax;
    out8[0] = MIMO_PLAN_TWO_STAGE; out8[1] = 1 + groups;
    out8[2] = gibbs ? 0 : 1;             // the (K, N) responsibility table goes through HBM
    out8[3] = gibbs ? 0 : groups;        // and is read once per statistics launch
    out8[4] = 1 + groups;                // Z: the E-step + every statistics launch
    out8[5] = gibbs ? 1 + groups : 0;
    const char* est = wide_estep_covers(a.K16, D, a.F16, gibbs) ? "wide_estep_kernel" : "estep_chunked_kernel";
    if (gibbs && label_stats_covers(K, D, ctx->structure)) {      // label-indexed statistics (as run_pass)
      const int ll = label_stage();
      out8[1] = 1 + ll; out8[4] = 1 + ll; out8[5] = 1 + ll;
      if (desc) snprintf(desc, dlen, "%s (label draw) + %s", est, lst);
    } else if (desc) {
      snprintf(desc, dlen, "%s (%s) + %s x %d", est, gibbs ? "label draw" : "softmax -> (K, N) table", wide ? "wide_stats_kernel" : "fused_kernel (statistics)", groups);
    }
  }
}

int mimo_plan(mimo_ctx* ctx, int K, int gibbs, int64_t* out8) {
  return guarded(ctx, [&]() -> int {
  int rc = bind(ctx); if (rc) return rc;
  if (!out8) return fail(ctx, MIMO_E_INVALID, "mimo_plan: out is NULL");
  if (!ctx->Z) return fail(ctx, MIMO_E_NODATA, "no data uploaded or attached");
  if (K < 1 || K > 256) return fail(ctx, MIMO_E_UNSUPPORTED, "K = %d outside [1, 256]", K);
  KernelArgs a;
  fill_args(ctx, K, &a);
  a.gibbs = gibbs ? 1 : 0;
  plan_route(ctx, a, K, gibbs, out8, nullptr, 0);
  switch (out8[0]) {
    case MIMO_PLAN_MID: out8[6] = gibbs ? mid_labels_grid(a, ctx->num_cu) : mid_grid(a, ctx->num_cu); break;
    case MIMO_PLAN_NARROW: out8[6] = narrow_grid(a, ctx->num_cu, ctx->F, use_narrow(ctx, K, gibbs != 0, true, true) - 1); break;
    case MIMO_PLAN_SMALL: out8[6] = small_grid(a, ctx->num_cu, kSrcEstep); break;
    case MIMO_PLAN_ROWWAVE_VI: case MIMO_PLAN_ROWWAVE: out8[6] = rowwave_grid(a, ctx->num_cu); break;
    case MIMO_PLAN_FUSED: out8[6] = fused_grid(a, ctx->num_cu, kSrcEstep); break;
    default:
      out8[6] = fused_grid(a, ctx->num_cu, kSrcEstep);
      if (wide_estep_covers(a.K16, ctx->D, a.F16, gibbs)) {                // as run_pass: two workgroups per CU
        const int64_t g2 = 2 * (int64_t)ctx->num_cu;
        out8[6] = g2 < a.ntiles ? g2 : (a.ntiles > 0 ? a.ntiles : 1);
      }
  }
  out8[7] = ctx->num_cu;
  return MIMO_OK;
  });
}

int mimo_plan_shape(int Dz, int K, int structure, int64_t N, int gibbs, int64_t* out8, char* desc, int desc_len) {
  return guarded(nullptr, [&]() -> int {
  if (!out8) return fail(nullptr, MIMO_E_INVALID, "mimo_plan_shape: out is NULL");
  if (Dz < 1 || Dz > kMaxD || K < 1 || K > 256 || N < 0 || structure < 0 || structure > 2)
    return fail(nullptr, MIMO_E_UNSUPPORTED, "mimo_plan_shape: Dz = %d, K = %d, structure = %d outside the library's range", Dz, K, structure);
  mimo_ctx ctx;                        // a description of the data, never a device context: no HIP call below
  alignas(16) static const double aligned_rows[2] = {0.0, 0.0};
  ctx.Z = aligned_rows; ctx.N = N; ctx.D = Dz; ctx.structure = structure;
  ctx.F = structure == MIMO_STRUCT_DIAG ? diag_feat_count(Dz) : structure == MIMO_STRUCT_LINEAR ? lin_feat_count(Dz) : feat_count(Dz);
  ctx.F16 = (ctx.F + 15) / 16 * 16;
  KernelArgs a;
  fill_args(&ctx, K, &a);
  a.gibbs = gibbs ? 1 : 0;
  if (desc && desc_len > 0) desc[0] = 0;
  plan_route(&ctx, a, K, gibbs, out8, desc, desc && desc_len > 0 ? (size_t)desc_len : 0);
  return MIMO_OK;
  });
}

}  // extern "C"
