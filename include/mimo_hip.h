/*
 * mimo_hip.h — C ABI of libmimo_hip.so, the MI355X (gfx950) E-step / sufficient-statistics
 * engine that drops in behind hanyas/mimo's NumPy hot path.
 *
 * The reference has no FFI seam: its hot path is a set of Python methods that turn the
 * (N,D) data array into (K,N) log-density tables, responsibilities / labels and
 * responsibility-weighted sufficient statistics.  Every entry point
 * below cites the reference method(s) it replaces (paths relative to the reference root).
 *
 * Canonical form evaluated on the device for every model in scope:
 *
 *     l[k,n] = c[k] + b[k,:]·z[n,:] − ½ z[n,:]ᵀ W[k,:,:] z[n,:]          (float64)
 *
 * with z the data row (GMM: x; mixture of linear-Gaussian experts: z=[x,y]) and (c,b,W) built on
 * the host from point estimates (Gibbs/EM) or posterior expectations (mean-field VI).
 *
 * Conventions
 *   - all floating point is IEEE float64; labels are int32; row counts int64.
 *   - data Z is row-major (N, Dz).  Per-component tables are K-major (K, N).
 *   - sufficient statistics are returned packed per component:
 *         S[k] = [ n_k , sum_n r_kn z_n (Dz) , sum_n r_kn z_n z_nᵀ (Dz×Dz, row-major, symmetric) ]
 *     i.e. K × (1 + Dz + Dz²) doubles  (mimo/distributions/gaussian.py:491-502,
 *     mimo/distributions/lingauss.py:306-322: yxT/xxT/yyT/n_k are blocks of this for z=[x,y]).
 *   - scalars[3] = { sum_n logsumexp_k l[k,n] , sum_n sum_k r_kn l[k,n] , −sum_n sum_k r_kn log r_kn }
 *     (mimo/mixtures/gmm.py:338-356).  scalars[0] alone gives the data + label ELBO terms; the split
 *     into scalars[1], scalars[2] costs extra float64 work per (datum, component) and is only
 *     computed with MIMO_F_ENTROPY_SPLIT or any MIMO_F_KEEP_* flag (otherwise both are NaN).
 *   - every function returns 0 on success or a negative MIMO_E_* code; the message is
 *     available from mimo_last_error().  No C++ exception crosses this boundary: every entry point runs
 *     inside a catch-all guard (std::bad_alloc -> MIMO_E_NOMEM, anything else -> MIMO_E_INTERNAL), and the
 *     error message lives in a fixed buffer, so reporting a failure never allocates.
 *   - pointers are HOST pointers unless MIMO_F_DEVICE_OUT is set in `flags`, in which case the
 *     *output* pointers (S, scalars) are DEVICE pointers, nothing is copied to the host and the
 *     call returns without synchronising the context's stream (used by the multi-GPU driver,
 *     which all-reduces S with RCCL before the host reads it).
 *   - one context per device, not thread-safe; the library owns all device memory behind it.
 */
#ifndef MIMO_HIP_H
#define MIMO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mimo_ctx mimo_ctx;

/* error codes */
#define MIMO_OK               0
#define MIMO_E_INVALID       -1   /* bad argument (shape, NULL, flags)            */
#define MIMO_E_HIP           -2   /* a HIP runtime call failed                    */
#define MIMO_E_NODATA        -3   /* no data uploaded / attached                  */
#define MIMO_E_UNSUPPORTED   -4   /* (K, Dz) outside the ranges the kernels cover */
#define MIMO_E_STATE         -5   /* requested buffer was never produced          */
#define MIMO_E_NOMEM         -6   /* host allocation failed inside the library    */
#define MIMO_E_INTERNAL      -7   /* any other C++ exception, caught at the boundary */

/* flags */
#define MIMO_F_KEEP_RESP      0x01  /* keep the (K,N) responsibility table on the device          */
#define MIMO_F_KEEP_LOGP      0x02  /* keep the (K,N) log-density table l[k,n] on the device       */
#define MIMO_F_KEEP_LSE       0x04  /* keep the (N,) per-datum log-normaliser on the device        */
#define MIMO_F_NO_STATS       0x08  /* skip the sufficient-statistics accumulation                 */
#define MIMO_F_DEVICE_OUT     0x10  /* S / scalars are device pointers; asynchronous               */
#define MIMO_F_DEVICE_IN      0x20  /* `resp` / `labels` / `u` inputs are device pointers          */
#define MIMO_F_ENTROPY_SPLIT  0x40  /* also produce scalars[1], scalars[2] (see above)             */
#define MIMO_F_ASYNC          0x80  /* enqueue only; fetch the host results with mimo_wait()       */
#define MIMO_F_DIAG_VAR       0x200 /* mimo_predict_flags: the second output is (N, dy) variances followed by (N, dy) standard deviations */
#define MIMO_F_WEIGHTS_RESIDENT 0x100 /* mimo_estep_weighted: reuse the row weights the previous weighted call uploaded
                                         (the hierarchical drivers pass the same vector every iteration: N doubles less
                                         over PCIe per call); `row_weights` is ignored                                */

/* ---- lifetime -------------------------------------------------------------------------- */

/* Create a context on HIP device `device`.  Replaces nothing in the reference (it has no
 * device state); the context plays the role of the Python objects' NumPy buffers. */
int mimo_create(mimo_ctx** out, int device);
int mimo_destroy(mimo_ctx* ctx);

/* Last error message for `ctx` (or the last context-less error when ctx == NULL). */
const char* mimo_last_error(const mimo_ctx* ctx);

/* Launch on an existing HIP stream instead of the context's own (non-blocking) stream.  Pass NULL to return to
 * the context's own stream — so the NULL stream itself cannot be selected: a caller that orders other work
 * (a collective, a copy) behind these kernels passes a stream it created (mimo_amd/sharded.py does). */
int mimo_set_stream(mimo_ctx* ctx, void* hip_stream);

/* ---- data ------------------------------------------------------------------------------ */

/* Copy the (N,Dz) row-major observation matrix to the device once; it stays resident across
 * sweeps.  Stands for the `obs` / `(x,y)` arrays every reference driver receives
 * (mimo/mixtures/gmm.py:207,261; mimo/mixtures/ilr.py:134,196 — there hstack((x,y))). */
int mimo_upload(mimo_ctx* ctx, const double* Z_host, int64_t N, int Dz);

/* Borrow an existing device buffer (e.g. a torch tensor) instead of copying. */
int mimo_attach(mimo_ctx* ctx, const double* Z_dev, int64_t N, int Dz);

/* Structure of the precision blocks W[k] of every later call on this context.
 *   MIMO_STRUCT_FULL (default): symmetric W, (Dz+1)(Dz+2)/2 features per row.
 *   MIMO_STRUCT_DIAG: diagonal W — the kernels contract only the 2 Dz + 1 features z_a^2, z_a, 1
 *     (3.3x fewer matrix instructions at Dz = 16) and cover Dz <= 32 in a single pass for K <= 64; the packed
 *     statistics keep their layout with zero off-diagonal second moments.  A W with a non-zero off-diagonal
 *     entry is rejected (MIMO_E_INVALID).
 *     Replaces: the diagonal-precision family — StackedGaussiansWithDiagonalPrecision.log_likelihood /
 *     weighted_statistics (mimo/distributions/gaussian.py:802-832) and
 *     StackedGaussiansWithNormalGammas.expected_log_likelihood (mimo/distributions/bayesian.py:441-455).
 *   MIMO_STRUCT_LINEAR: every W[k] is the SAME matrix (tied covariance).  The term -1/2 z'W z is then common
 *     to all components: it cancels in the softmax and in the label draw, and its sums are data constants,
 *       sum_n lse_n = sum_n lse_n(linear part) - 1/2 tr(W XX),   sum_k sum_n r_kn z z' = XX = sum_n z_n z_n',
 *     so the kernels contract only the Dz + 1 features z_a, 1 (5x fewer matrix instructions at Dz = 16).
 *     Returned: n_k and sum r z in the packed block (second moments zero), scalars WITHOUT the -1/2 tr(W XX)
 *     term; the caller owns XX (one MIMO_STRUCT_FULL call with K = 1 and unit weights per data set).  Requests
 *     for the log-density / log-normaliser tables need the full structure.  W[k] != W[0] is rejected.
 *     Replaces: the same methods for TiedGaussiansWithPrecision / TiedGaussiansWithNormalWisharts
 *     (gaussian.py:545-572, bayesian.py:326-340, composite.py:259-283) and the hierarchical block
 *     (bayesian.py:734-755), whose updates only ever use sum_k of the second-moment blocks. */
#define MIMO_STRUCT_FULL 0
#define MIMO_STRUCT_DIAG 1
#define MIMO_STRUCT_LINEAR 2
int mimo_set_structure(mimo_ctx* ctx, int structure);

/* Global index of local row 0; only enters the Philox counter so that labels drawn by a sharded
 * run do not depend on the number of shards.  Default 0. */
int mimo_set_row_offset(mimo_ctx* ctx, int64_t row0);

/* ---- the hot path ---------------------------------------------------------------------- */

/* Fused mean-field / EM E-step: log-densities, softmax over k, weighted sufficient statistics
 * and ELBO scalars in ONE pass over Z.
 * Replaces: StackedGaussiansWithNormalWisharts.expected_log_likelihood
 *   (mimo/distributions/bayesian.py:287-301) or StackedGaussiansWithPrecision.log_likelihood
 *   (mimo/distributions/gaussian.py:510-521) [+ the linear-Gaussian counterparts
 *   bayesian.py:933-947, lingauss.py:330-345], the softmax of
 *   expected_responsibilities / responsibilities (mimo/mixtures/gmm.py:72-75,256-259),
 *   weighted_statistics (gaussian.py:491-502, lingauss.py:306-322, categorical.py:41-43) and the
 *   data/label ELBO terms (gmm.py:338-356).
 * c (K), b (K,Dz), W (K,Dz,Dz).  S: K×(1+Dz+Dz²) or NULL with MIMO_F_NO_STATS; scalars: 3 or NULL.
 * c[k] = -inf (a component whose weight is exactly 0: log(probs) in gmm.py:72) is allowed and gives r_kn = 0
 * (it enters the kernels as -1e300; an l table kept with MIMO_F_KEEP_LOGP holds about -1e300 there, not -inf);
 * NaN / +inf anywhere in c, b, W is MIMO_E_INVALID.  The same holds for every entry point that takes (c, b, W). */
int mimo_estep(mimo_ctx* ctx, const double* c, const double* b, const double* W, int K,
               int flags, double* S, double* scalars);

/* The same pass with a weight per data row: the statistics are those of r_kn * row_weights[n], the tables
 * and the ELBO scalars those of the unweighted r_kn.
 * Replaces: the `weights` argument of the hierarchical drivers — resp * weights feeds the update while the
 *   bound uses resp (mimo/mixtures/hgmm.py:199-207), and an outer mixture hands its responsibilities
 *   resp[m, :] to inner mixture m as such weights (hgmm.py:407-414, 457-465).
 * row_weights: N doubles (host, or device with MIMO_F_DEVICE_IN).  Shapes on the two-stage path
 * (Dz > 16, or K > 64 with Dz > 9) return MIMO_E_UNSUPPORTED: there the weights go in as a table
 * (mimo_estep with MIMO_F_KEEP_RESP, then mimo_weighted_stats). */
int mimo_estep_weighted(mimo_ctx* ctx, const double* c, const double* b, const double* W, int K,
                        const double* row_weights, int flags, double* S, double* scalars);

/* Completes a call issued with MIMO_F_ASYNC: waits for the context's stream and copies the packed
 * statistics / scalars of that call to the host pointers (either may be NULL).  Lets the host overlap
 * its own O(K D^3) work (the ELBO's prior terms, gmm.py:360-361) with the pass over the data. */
int mimo_wait(mimo_ctx* ctx, double* S, double* scalars);

/* Fused Gibbs label step: log-densities, categorical draw per datum by inverse CDF, and the
 * sufficient statistics of the labels just drawn (what the next sweep's resample_components /
 * resample_gating consume) in ONE pass.
 * Replaces: resample_labels (mimo/mixtures/gmm.py:227-230, ilr.py:161-164) =
 *   log_complete_likelihood + sample_discrete_from_log (mimo/utils/stats.py:8-21), then
 *   one_hot (mimo/utils/data.py:160-169) + weighted_statistics + Categorical.statistics
 *   (categorical.py:35-37) of the following sweep.
 * Uniforms: if `u` != NULL it holds N uniforms in [0,1) (reference-exact mode: the values
 *   numpy.random.random((1,N)) returned); otherwise datum n uses the in-kernel counter-based
 *   Philox4x32-10 stream keyed by `seed` with counter (row0+n, sweep).
 * labels_out: N int32 on the host or NULL (labels always stay on the device as well). */
int mimo_gibbs_labels(mimo_ctx* ctx, const double* c, const double* b, const double* W, int K,
                      uint64_t seed, uint64_t sweep, const double* u,
                      int flags, int32_t* labels_out, double* S);

/* Sufficient statistics for arbitrary weights resp (K,N), K-major.
 * Replaces: weighted_statistics(data, weights) (gaussian.py:491-502, lingauss.py:306-322,
 * categorical.py:41-43) when the weights do not come from mimo_estep (random initial
 * responsibilities gmm.py:265-267, EM with user weights gmm.py:93). */
int mimo_weighted_stats(mimo_ctx* ctx, const double* resp, int K, int flags, double* S);

/* Sufficient statistics of hard labels without materialising one_hot(labels,K).
 * Replaces: one_hot + weighted_statistics (gmm.py:235-237, data.py:160-169) and
 * Categorical.statistics = bincount (categorical.py:35-37).  labels == NULL uses the labels the
 * last mimo_gibbs_labels left on the device. */
int mimo_label_stats(mimo_ctx* ctx, const int32_t* labels, int K, int flags, double* S);

/* Categorical draw per column of a (K, N) table of (unnormalised) log-probabilities, K-major:
 *   label_n = #{k : u_n cum_K > cum_k},  cum = cumsum_k exp(logp[k,n] - max_k),  lognorm_n = logsumexp_k logp[k,n].
 * Replaces: mimo.utils.stats.sample_discrete_from_log (mimo/utils/stats.py:8-21; imported by the drivers at
 *   mimo/mixtures/gmm.py:9-10, ilr.py:9-10) for a table the caller already holds — the fused label step
 *   (mimo_gibbs_labels) never materialises that table and is what the drivers of this library use.
 * logp: host table, or device with MIMO_F_DEVICE_IN, or NULL = the log-density table a call with MIMO_F_KEEP_LOGP left
 *   on the device (then K and N must be its shape).  u: N uniforms in [0,1) (the values the reference's single
 *   numpy.random.random(size=(1,N)) call returned) or NULL for the Philox stream (seed; counter (row0 + n, sweep)).
 * labels_out: N int32 (host).  lognorms_out: N doubles (host) or NULL. */
int mimo_sample_from_log(mimo_ctx* ctx, const double* logp, int K, int64_t N, const double* u, uint64_t seed,
                         uint64_t sweep, int flags, int32_t* labels_out, double* lognorms_out);

/* Statistics of RANDOM initial responsibilities generated on the device: r[k,n] = v_kn / sum_j v_jn, v_kn the
 * Philox4x32-10 uniform of key `seed`, counter (row0 + n, k) — the start of the drivers with randomize=True
 * (mimo/mixtures/gmm.py:265-267, ilr.py:200-202: resp = rand(K,N); resp /= resp.sum(0); weighted_statistics)
 * without K N host uniforms going over PCIe (5 GB at N = 1e7, K = 64).  The table stays resident (mimo_get_resp).
 * A stream of its own: numpy's generator cannot be continued on the device, so seeded runs that must reproduce the
 * reference's trace keep the host draw + mimo_weighted_stats. */
int mimo_random_resp_stats(mimo_ctx* ctx, int K, uint64_t seed, int flags, double* S);

/* -sum_{k,n} t log t of a (K,N) table (entries <= 0 contribute 0, like nansum).
 * Replaces: the entropy term of variational_lowerbound_labels for caller-supplied responsibilities
 * (mimo/mixtures/gmm.py:353-355, ilr.py:310-312).  table == NULL uses the resident resp table. */
int mimo_table_entropy(mimo_ctx* ctx, const double* table, int64_t count, int flags, double* out);

/* Posterior-predictive mixture moments for every resident row x_n (Z = the inputs, Dz = dx).
 * Replaces: BayesianMixtureOfLinearGaussians.meanfield_prediction and its helpers
 * (mimo/mixtures/ilr.py:339-372, 374-430) together with
 * StackedGaussiansWithNormalWisharts.log_posterior_predictive_gaussian (bayesian.py:303-313) and
 * StackedLinearGaussiansWithMatrixNormalWisharts.posterior_predictive_gaussian (bayesian.py:949-962).
 *   (c, b, W)[K]   canonical form of  log E[pi_k] + log N(x; basis predictive)   over x
 *   M  (K, dy, dc) posterior mean regression matrices, dc = dx + affine (x~ = [x, 1])
 *   Q  (K, dc, dc) cs_kn = 1 + x~' Q_k x~           (Q_k = K_k^-1)
 *   Cc (K, dy, dy) expert predictive covariance at cs = 1:  V_kn = cs_kn Cc_k  (= (df_k psi_k)^-1)
 *   mode 0: mixture moments  mu = sum_k w m_k, covar = sum_k w (V + m m') - mu mu'   (ilr.py:364-372)
 *   mode 1: moments of the arg-max-weight component                                   (ilr.py:395-398)
 *   y (N, dy), P (K, dy, dy) = Cc_k^-1, ld (K) = logdet P_k  -> nlpd (N) =
 *       -logsumexp_k [ log N(y_n; m_kn, (P_k / cs_kn)^-1) + log(w_kn + tiny) ]        (ilr.py:405-409);
 *       all four NULL to skip.  Outputs on the host: mu (N, dy), covar (N, dy, dy).
 * dy <= 8; MIMO_E_UNSUPPORTED otherwise. */
int mimo_predict(mimo_ctx* ctx, const double* c, const double* b, const double* W, int K,
                 const double* M, const double* Q, const double* Cc, int dy, int affine, int mode,
                 const double* y, const double* P, const double* ld,
                 double* mu, double* covar, double* nlpd);
/* The same with flags: MIMO_F_DIAG_VAR — `covar` receives 2 N dy doubles: the variances (N, dy) (the diagonal of the covariance:
 * what the reference's callers read, ilr.py:411-417) followed by the standard deviations (N, dy); MIMO_F_DEVICE_IN — y is a device pointer; MIMO_F_DEVICE_OUT — mu, covar (and nlpd) are device
 * pointers the kernel writes directly, and the call returns without waiting (the results are ordered on the context's
 * stream like every other launch: mimo_set_stream / the caller's next stream operation).  With both flags nothing but
 * the K parameter blocks crosses PCIe: at N = 4e6, dx = dy = 1 the host-array form spends 120 of its 125 ms there. */
int mimo_predict_flags(mimo_ctx* ctx, const double* c, const double* b, const double* W, int K,
                       const double* M, const double* Q, const double* Cc, int dy, int affine, int mode,
                       const double* y, const double* P, const double* ld,
                       double* mu, double* covar, double* nlpd, int flags);

/* ---- rows with missing values ------------------------------------------------------------------
 * A data row that holds a NaN is treated the way the reference's Gaussian family treats it: it is LEFT OUT of every
 * sufficient statistic (mimo/distributions/gaussian.py:493-494: idx = ~isnan(data).any(axis=1); also :468, :650, :796)
 * and its log-density is the normaliser-only value, i.e. the canonical form at z = 0 (gaussian.py:512-520:
 * nan_to_num, then the data-dependent part of those rows is set to 0) — tables, labels and the ELBO scalars carry it
 * like any other row.  mimo_upload / mimo_attach find such rows once (one pass over the device copy; a borrowed buffer
 * is copied before it is touched), every later call on the data set applies the row mask.
 * The reference's GATING update counts the labels / responsibilities of those rows (bincount(labels), resp.sum(1)
 * over all rows) while its component update drops them; the packed block returned here is the component one, and
 * mimo_nan_info gives what the gating update needs on top:
 *   n_bad          number of rows with a NaN (0: nothing below applies)
 *   row_mask_out   N doubles, 1 = complete row (NULL to skip)
 *   label_counts   K int64: labels the last label pass (mimo_gibbs_labels / mimo_label_stats) put on NaN rows, per
 *                  component (NULL to skip).  For a softmax pass every NaN row has the responsibilities
 *                  softmax_k(c_k): n_bad times that vector is the gating part, computed by the caller.
 * Linear-Gaussian experts: the rule applies to the joint row z = [x, y] (a NaN anywhere drops the row from the
 * statistics, lingauss.py:103-104); the reference's log-density only zeroes rows where x AND y hold a NaN
 * (lingauss.py:150-151) and evaluates the others on element-wise nan_to_num'ed values: the host mirror puts those values in
 * place of the library's whole-row value for the few rows concerned (mimo_amd/mixtures/ilr.py, nan_rows_table). */
int mimo_nan_info(mimo_ctx* ctx, int64_t* n_bad, double* row_mask_out, int K, int64_t* label_counts);

/* ---- sharding over the GPUs of a node (one process per GPU) ----------------------------------
 * The path shards over rows: every per-datum quantity is local, the only exchange per pass is the sum of the packed
 * statistic block [K (1 + Dz + Dz^2)] + 3 scalars over the ranks (0.14 MB at K = 64, Dz = 16).
 * With a communicator attached, every entry point that returns statistics / scalars (mimo_estep, mimo_estep_weighted,
 * mimo_gibbs_labels, mimo_weighted_stats, mimo_label_stats, mimo_random_resp_stats, mimo_wait) returns them SUMMED OVER
 * THE RANKS: one RCCL all-reduce(sum, float64) on the context's stream behind the kernels, before the copy to the
 * host (or in place in the caller's device buffer with MIMO_F_DEVICE_OUT).  Labels and tables stay with their rank.
 * Every rank makes the same calls in the same order (collective semantics); mimo_set_row_offset gives each rank the
 * global index of its first row so that the Philox label stream does not depend on the number of ranks.
 * The reference has no counterpart (single process, NumPy).  mimo_amd/sharded.py is the same step through
 * torch.distributed for Python hosts; these entry points are for hosts without it.
 *   rank 0: mimo_comm_unique_id(id)  ->  the application hands the 128 bytes to the other ranks  ->
 *   every rank: mimo_comm_init(ctx, id, rank, world)
 * RCCL is opened at run time; MIMO_E_UNSUPPORTED if librccl cannot be loaded. */
int mimo_comm_unique_id(char* id128);
int mimo_comm_init(mimo_ctx* ctx, const char* id128, int rank, int world);
int mimo_comm_destroy(mimo_ctx* ctx);

/* ---- copy-outs of device-resident tables ----------------------------------------------- */
int mimo_get_resp(mimo_ctx* ctx, double* resp_host /* K×N */);
/* The columns `cols[0 .. ncols)` (row indices of the data) of the resident responsibility table: out (K, ncols) row-major.  What a
 * host-side correction over a few rows needs (the share of the rows with NaN in the gating counts: categorical.py:35-46 counts
 * them, gaussian.py:493-494 drops them) without copying the (K, N) table.  MIMO_E_STATE without a resident table. */
int mimo_get_resp_columns(mimo_ctx* ctx, const int64_t* cols, int64_t ncols, double* out);
int mimo_get_logp(mimo_ctx* ctx, double* logp_host /* K×N */);
int mimo_get_lse(mimo_ctx* ctx, double* lse_host /* N */);
int mimo_get_labels(mimo_ctx* ctx, int32_t* labels_host /* N */);

/* ---- host-side conjugate algebra of the variational sweep (no GPU calls) ------------------
 * The consumer of the statistics block and producer of the next launch's parameters: K small dense
 * problems per sweep, batched in one call (threads over k when K D^3 is large).  Inputs are the
 * posterior NATURAL parameters (prior + statistics).  Return MIMO_E_INVALID if a block is not
 * positive definite (the caller then takes the NumPy route, which raises like the reference). */

/* Normal-Wishart blocks.  Replaces per sweep: NormalWishart.nat_to_std (composite.py:67-72),
 * expected_statistics (composite.py:106-118, wishart.py:139-143) and the canonical form of
 * StackedGaussiansWithNormalWisharts.expected_log_likelihood (bayesian.py:287-301).
 *   in : a (K,D) = kappa m, b (K) = kappa, c (K,D,D) = psi^-1 + kappa m m', d (K) = nu - D
 *   out: mus (K,D), psis (K,D,D), nus (K), half_logdet_psi (K) = sum log diag chol psi,
 *        (cc, bb, W) = canonical expected log-density WITHOUT the gating term,
 *        E2 (K) = E[-1/2 mu'Lambda mu], E4 (K) = E[1/2 logdet Lambda]   (E1 = bb, E3 = -W/2). */
int mimo_host_nw_vi(int K, int D, const double* a, const double* b, const double* c, const double* d,
                    double* mus, double* psis, double* nus, double* half_logdet_psi,
                    double* cc, double* bb, double* W, double* E2, double* E4);

/* The same for TIED Normal-Wishart blocks (one Wishart factor shared by all k): replaces
 * TiedNormalWishart.nat_to_std (composite.py:273-283: psi = inv(mean_k(c_k - kappa_k m_k m_k')), nu = mean_k(d_k + D))
 * and the std_to_nat that the reference recomputes on every read of the tied natural parameters
 * (composite.py:166-172): nat_c (K,D,D) = psi^-1 + kappa_k m_k m_k' of the POOLED psi.  psis, nus,
 * half_logdet_psi, W, E4 hold K copies of the shared value. */
int mimo_host_nw_vi_tied(int K, int D, const double* a, const double* b, const double* c, const double* d,
                         double* mus, double* psis, double* nus, double* half_logdet_psi, double* nat_c,
                         double* cc, double* bb, double* W, double* E2, double* E4);

/* The Normal-Wishart blocks' term of the variational lower bound, entropy(q_k) - cross_entropy(q_k, p_k) per k
 * (bayesian.py:258-265 with composite.py:95-98,120-128 and wishart.py:129-132), from the quantities the two calls
 * above return: (qa..qd) the posterior's natural parameters, (pa..pd) the prior's, prior_logZ (K) the prior's
 * log-partition, nus / half_logdet_psi / E1 = bb / E2 / W / E4 of the posterior.  out (K). */
int mimo_host_nw_vlb(int K, int D, const double* qa, const double* qb, const double* qc, const double* qd,
                     const double* pa, const double* pb, const double* pc, const double* pd, const double* prior_logZ,
                     const double* nus, const double* half_logdet_psi, const double* E1, const double* E2,
                     const double* W, const double* E4, double* out);

/* One mean-field sweep's host algebra for a mixture of Gaussians with Normal-Wishart blocks (tied != 0: one shared
 * Wishart factor) and Dirichlet gating, i.e. what BayesianMixtureOfGaussians.meanfield_coordinate_descent runs between two
 * data passes (gmm.py:275-285) in TWO calls: _sweep before the next pass is launched (meanfield_update of components and
 * gating, bayesian.py:78-83,225-230; the expected log-densities, gmm.py:244-254), _bound while it runs (the bound's prior
 * terms, bayesian.py:93-96,258-265; dirichlet.py:78-97).
 *   _sweep in : alpha0 (K) prior concentrations, counts (K) = sum_n r_kn; (pa..pd) the components' prior natural
 *               parameters; (sx, sn, sxx) = the statistics block of the last pass (K,D) (K) (K,D,D)
 *   _sweep out: alpha (K) posterior concentrations; (qa..qd) the posterior natural parameters as assigned; the outputs of
 *               mimo_host_nw_vi / _tied (nat_c may be NULL when tied == 0); e_log_pi (K) = E[log pi_k];
 *               c_total (K) = cc + e_log_pi: (c_total, bb, W) is what the next mimo_estep takes
 *   _bound    : the same arrays back, prior_logZ (K) the components' prior log-partition;
 *               vlb[0] = gating term, vlb[1] = sum over k of the component terms. */
int mimo_host_gmm_vi_sweep(int K, int D, int tied, const double* alpha0, const double* counts,
                           const double* pa, const double* pb, const double* pc, const double* pd,
                           const double* sx, const double* sn, const double* sxx,
                           double* alpha, double* qa, double* qb, double* qc, double* qd,
                           double* mus, double* psis, double* nus, double* half_logdet_psi, double* nat_c,
                           double* cc, double* bb, double* W, double* E2, double* E4, double* e_log_pi, double* c_total);
int mimo_host_gmm_vi_bound(int K, int D, int tied, const double* alpha0, const double* alpha, const double* e_log_pi,
                           const double* pa, const double* pb, const double* pc, const double* pd, const double* prior_logZ,
                           const double* qa, const double* qb, const double* qc, const double* qd,
                           const double* mus, const double* nus, const double* half_logdet_psi, const double* nat_c,
                           const double* bb, const double* E2, const double* W, const double* E4, double* vlb);

/* K blocks of variates from numpy.random's LEGACY stream (RandomState over MT19937), bit for bit what the reference's
 * per-component calls consume and return: per block k, n_before x normal(), then standard_gamma(shapes[k][i]) for i < n_gamma
 * (chisquare(df) = 2 standard_gamma(df / 2), gamma(a, scale) = scale standard_gamma(a): the caller scales), then n_after x
 * normal().  Replaces the Python loops over k around wishart.py:72-92 (Bartlett factors), composite.py:82-86,347-351,612-617
 * (the mean / matrix draw that follows) and gamma.py:53-55.  The generator state travels in and out:
 * (mt_key[624], *mt_pos, *has_gauss, *gauss) = numpy.random.get_state()[1:], to be handed back with set_state. */
int mimo_host_legacy_draws(uint32_t* mt_key, int* mt_pos, int* has_gauss, double* gauss, int K, int n_before,
                           int n_gamma, int n_after, const double* shapes, double* before, double* gammas,
                           double* after);

/* The mean-field update of the hierarchical block (K Gaussians whose means share a Normal-Wishart hyper-posterior): nb_iter rounds of
 * "component means given the hyper-posterior mean, pooled hyper-posterior given the means" —
 * TiedGaussiansWithHierarchicalNormalWisharts.meanfield_update, bayesian.py:661-689 with the pooled block of :670-682.
 *   in : kap (K) the prior's kappas; the hyper-prior (m0 (D), kappa0, psi0_inv (D,D) = inv(psi0), nu0); xk (K,D), nk (K),
 *        sxx_sum (D,D) = sum_k sum_n r_kn x x' — the statistics of the last pass; mu_q (D) the hyper-posterior mean to start from
 *   out: mu_q (D), kappa_q, psi_q (D,D), nu_q — the hyper-posterior; post_mus (K,D), post_kappas (K) — the components' posterior.
 * MIMO_E_INVALID if a pooled block is not positive definite (the caller then takes the NumPy route). */
int mimo_host_hier_vi(int K, int D, int nb_iter, const double* kap, const double* m0, double kappa0, const double* psi0_inv,
                      double nu0, const double* xk, const double* nk, const double* sxx_sum, double* mu_q, double* post_mus,
                      double* post_kappas, double* kappa_q, double* psi_q, double* nu_q);

/* The same draws on numpy's generator IN PLACE: mt_key / mt_pos point into the bit generator's own state
 * (numpy.random.mtrand._rand._bit_generator.ctypes.state_address: uint32 key[624], int pos), so nothing is copied in or out
 * (get_state / set_state cost 0.1 ms together — numpy converts the key element by element).  The cached second gaussian lives in
 * RandomState, out of reach: the caller passes what it found there (has_gauss, gauss — after emptying the cache), and when the
 * draws end with a gaussian cached, the call writes the state it reached to final_state (key[624], pos), leaves the key / position
 * where the PAIR behind that gaussian was begun and sets *redraw: one numpy.random.standard_normal() by the caller reproduces the
 * pair, returns the half that was used and caches the other; the caller then copies final_state over the generator's state (uniforms
 * drawn after the pair) — numpy ends in exactly the state the per-component calls would have left
 * (mimo_amd/distributions/wishart.py: legacy_draws). */
int mimo_host_legacy_draws_inplace(uint32_t* mt_key, int* mt_pos, int has_gauss, double gauss, int K, int n_before,
                                   int n_gamma, int n_after, const double* shapes, double* before, double* gammas,
                                   double* after, int* redraw, uint32_t* final_state);

/* random.sample(range(n), k) of CPython's `random` module (Lib/random.py: both branches of `sample`, _randbelow_with_getrandbits over
 * the MT19937 stream) — the minibatch indices of the SVI drivers (mimo/utils/data.py:9-12): the same list and the same generator state,
 * ~10 ns per index instead of ~0.4 us through three Python frames.  (mt_key[624], *mt_pos) = random.getstate()[1], in and out;
 * use_pool = CPython's choice between its two branches (n <= 21 + 4^ceil(log4(3k)) for k > 5, computed by the caller in the same
 * floating-point arithmetic); n < 2^32. */
int mimo_host_py_sample(uint32_t* mt_key, int* mt_pos, int64_t n, int64_t k, int use_pool, int64_t* out);

/* Matrix-Normal-Wishart blocks (experts y | x).  Replaces per sweep: MatrixNormalWishart.nat_to_std
 * (composite.py:594-599), expected_statistics (composite.py:635-647) and the canonical form of
 * StackedLinearGaussiansWithMatrixNormalWisharts.expected_log_likelihood (bayesian.py:933-947).
 *   in : a (K,dy,dc) = M K, b (K,dc,dc) = K, c (K,dy,dy) = psi^-1 + M K M', d (K) = nu - dy - 1 + dc
 *   out: Ms (K,dy,dc), psis (K,dy,dy), nus (K), half_logdet_psi (K), Kinv (K,dc,dc),
 *        (cc, bb, W) over z = [x, y] (Dz = dc - affine + dy), E1 (K,dy,dc), E2 (K,dc,dc), E4 (K). */
int mimo_host_mnw_vi(int K, int dy, int dc, int affine, const double* a, const double* b, const double* c,
                     const double* d, double* Ms, double* psis, double* nus, double* half_logdet_psi,
                     double* Kinv, double* cc, double* bb, double* W, double* E1, double* E2, double* E4);

/* Gibbs draw of K Normal-Wishart posteriors and the canonical form of the drawn Gaussians, in one call.  Replaces
 * per sweep: StackedNormalWisharts.rvs (composite.py:82-86) = K x Wishart.rvs (Bartlett, wishart.py:72-92) +
 * K x GaussianWithPrecision.rvs (gaussian.py:311-313, two more Cholesky factorisations each), and the log-partition /
 * canonical form of the refreshed likelihood (gaussian.py:352-354, 510-521).  The random variates come from the
 * caller (any generator), so this routine is deterministic:
 *   in : mus (K,D), kappas (K), psis (K,D,D) — standard parameters of the posterior;
 *        z (K, D(D-1)/2) standard normals = strict lower triangle of the Bartlett factor in the order of
 *        numpy.tril_indices(D, -1); g (K,D) = sqrt of chi-square(nu_k - i) draws (its diagonal); eps (K,D) standard normals
 *   out: mu (K,D), lmbda (K,D,D) — the draw: Lambda = T T' with T = chol(psi) A, mu = m + (sqrt(kappa) T)^-T eps;
 *        c (K), b (K,D) — log N(x; mu, Lambda^-1) = c + b.x - 1/2 x'Lambda x  (W = lmbda).
 * MIMO_E_INVALID if a psi_k is not positive definite. */
int mimo_host_nw_gibbs(int K, int D, const double* mus, const double* kappas, const double* psis,
                       const double* z, const double* g, const double* eps,
                       double* out_mu, double* out_lmbda, double* out_c, double* out_b);

/* Position-dependent checksum over every byte of a host buffer.  With w_i the 8-byte little-endian words (the last nbytes % 8
 * bytes zero-extended), m_i = w_i ^ (w_i >> 32) and nw their number:  out[0] = sum_i m_i,  out[1] = sum_i (nw - i) m_i  (mod 2^64).
 * Memory-bandwidth bound, threaded above 8 MB.  What engine.bind() / the row-weight residency of the host mirror key on, so
 * that an in-place edit of a caller's array between two calls is seen (the reference re-reads its arguments on every call:
 * mimo/mixtures/gmm.py:62-75): any edit of one word changes both sums, a swap of two unequal words (a row shuffle) changes
 * out[1].  mimo_data_checksum returns the same function of the rows mimo_upload copied, computed on the device. */
int mimo_host_checksum(const void* data, size_t nbytes, uint64_t out[2]);

/* digamma used by the two routines above (recurrence + asymptotic series), for tests. */
double mimo_host_digamma(double x);

/* ---- introspection --------------------------------------------------------------------- */

/* Uniform the in-kernel Philox stream gives datum `row` at `sweep` (host-side mirror, used by
 * tests to feed the same uniforms to the oracle). */
double mimo_philox_uniform(uint64_t seed, uint64_t row, uint64_t sweep);

/* Accumulated device time (ms, HIP events on the launch stream) and launch count of the
 * dominant fused kernel since the last reset; enable with mimo_profile(ctx, 1). */
int mimo_profile(mimo_ctx* ctx, int enable);
int mimo_profile_read(mimo_ctx* ctx, double* kernel_ms, int64_t* launches, int reset);

/* Median shader clock (MHz) of a short float64 VALU loop on every SIMD, from the in-kernel counters (s_memtime against the
 * constant 100 MHz s_memrealtime): what the device runs at under this kind of load, right now.  bench.py reports it after
 * its sustained leg. */
int mimo_shader_clock_mhz(mimo_ctx* ctx, double* mhz);

/* Per-kernel breakdown of the same measurement: one text line "name<TAB>total_ms<TAB>launches" per kernel of the
 * passes since the last reset (mimo_profile_read with reset = 1 clears it). */
int mimo_profile_kernels(mimo_ctx* ctx, char* buf, int len);

/* How a pass over the resident data with K components (softmax pass: gibbs = 0, label pass: gibbs = 1) will run:
 *   out8[0] kind (MIMO_PLAN_*), out8[1] kernels per pass (without the two small reduction kernels),
 *   out8[2] 1 if the (K, N) responsibility table goes through HBM, out8[3] how many times it is read back,
 *   out8[4] reads of the data Z per pass, out8[5] writes + reads of the (N,) labels per pass,
 *   out8[6] workgroups of the dominant kernel, out8[7] compute units of the device.
 * bench.py prices its HBM roofline block from this instead of assuming a single fused pass. */
#define MIMO_PLAN_FUSED      1   /* one fused tile kernel (MFMA): log-densities, softmax / draw, statistics        */
#define MIMO_PLAN_TWO_STAGE  2   /* chunked E-step + statistics launches per feature column group (Dz > 16, ...)    */
#define MIMO_PLAN_SMALL      3   /* small-shape VALU kernel (Dz <= 4, K <= 32): bound by HBM                        */
#define MIMO_PLAN_ROWWAVE    4   /* label pass: row-owner label kernel + label-indexed statistics kernel            */
#define MIMO_PLAN_ROWWAVE_VI 5   /* softmax pass at K <= 64, Dz <= 9: row-owner kernel for both matrix products     */
#define MIMO_PLAN_NARROW     6   /* Dz <= 4 with 32 < K <= 128: 4x4x4 matrix-instruction kernels (+ label statistics) */
#define MIMO_PLAN_MID        7   /* softmax pass of the mid shapes (K <= 128; mimo_mid.hip): row-owner E-step + column-owner statistics waves */
int mimo_plan(mimo_ctx* ctx, int K, int gibbs, int64_t* out8);

/* The same routing decision for a SHAPE — rows of Dz columns, N of them, K components, structure MIMO_STRUCT_*, softmax pass
 * (gibbs = 0) or label pass — without a context, data or a device: out8[0 .. 5] as mimo_plan (out8[6], the grid, stays 0:
 * it depends on the device's occupancy), and, if desc is not NULL, a one-line description of the kernels of the pass.  Host-only
 * (runs on a machine without a GPU): what ROUTING.md is generated from and what tests/test_routing.py checks it against, so the
 * routing constants are tested, not narrated.  No reference counterpart. */
int mimo_plan_shape(int Dz, int K, int structure, int64_t N, int gibbs, int64_t* out8, char* desc, int desc_len);

/* Checksum of the rows as mimo_upload received them (before rows with NaN are zeroed in the library's copy): the function
 * mimo_host_checksum defines, computed on the device inside the NaN scan of the upload (one read of Z either way).  A caller that
 * keeps the host array can compare the two to learn whether the array was edited since the upload (engine.bind() does, on a
 * helper thread under the first kernel of a call).  MIMO_E_STATE after mimo_attach (borrowed device rows: nothing was copied)
 * or without data.  Reference counterpart: none — it re-reads the caller's array on every call (mimo/mixtures/gmm.py:261). */
int mimo_data_checksum(mimo_ctx* ctx, uint64_t out[2]);

/* Test / tuning hook: overrides a launch-geometry value for later calls.  Keys:
 *   "num_cu"       (per context) the compute-unit count every persistent grid is sized from; 0 restores the device's own.
 *                  A small value makes every workgroup of every kernel family walk many tiles / steps / ranges at test sizes.
 *   "sorted_range" (process-wide) cap on the 256-row tiles per range of label_stats_sorted_kernel (default and maximum 80;
 *                  0 restores it): with a low cap a workgroup takes several ranges ("first range writes, later ranges add").
 *   "mid_min_d", "mid_narrow_k" (process-wide) 0 / 0: the mid kernels (mimo_mid.hip) run where they measured fastest.  Otherwise
 *                  they take every shape they exist for with Dz >= mid_min_d (default 5; > 32: the route is off), and the narrow
 *                  kernels keep the shapes both cover with K < mid_narrow_k (default 33).
 *   "mid_labels_min_d" (process-wide) smallest Dz whose label pass (K <= 48) runs on the mid kernel's label mode; 0: where it measured ahead.
 *   "narrow_big_vi" (process-wide) largest K (129 .. 256) whose softmax pass over at most two contraction steps runs on the narrow
 *                  kernels of mimo_narrow_big.hip; 0: where they measured ahead (256 for one step, 192 for two).
 * Results do not depend on either value beyond the documented summation order of the partial blocks.  MIMO_E_INVALID for
 * an unknown key or a value out of range.  The reference has no counterpart (launch geometry is ours). */
int mimo_tune(mimo_ctx* ctx, const char* key, int64_t value);

/* Test hook for the no-exception contract: throws, INSIDE the guarded boundary, kind 1: std::bad_alloc,
 * 2: std::runtime_error, 3: a non-std exception; returns the code the guard maps it to (MIMO_E_NOMEM,
 * MIMO_E_INTERNAL, MIMO_E_INTERNAL) with the message in mimo_last_error(ctx) (ctx may be NULL).  kind 0: MIMO_OK.
 * mimo_host_debug_fault does the same for the host-only entry points (mimo_host_*), kind 4 additionally makes
 * a worker thread of the batched routines fail to start (the call then finishes on the threads that did). */
int mimo_debug_fault(mimo_ctx* ctx, int kind);
int mimo_host_debug_fault(int kind);

/* Library version string. */
const char* mimo_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MIMO_HIP_H */
