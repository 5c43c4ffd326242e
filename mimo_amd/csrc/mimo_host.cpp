// Host-side conjugate algebra of the variational sweep, batched over the K components (float64, no GPU
// calls).  The fused kernel leaves one K x (1 + Dz + Dz^2) statistics block per sweep; what stands between
// that block and the next launch is K small dense problems (natural -> standard parameters, a Cholesky
// inverse, digamma sums, the canonical (c, b, W) of the expected log-density).  In NumPy that is ~40
// array calls and two LAPACK gufuncs (0.36 ms at K=64, D=16; 1.8 ms at K=128, D=32) on the critical
// path of every sweep; here it is one call.  Declared in include/mimo_hip.h.
//
// Reference semantics: mimo/distributions/composite.py:50-72,106-118 (Normal-Wishart),
// :577-599,635-647 (Matrix-Normal-Wishart), wishart.py:139-143, bayesian.py:287-301,933-947.
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cmath>
#include <pthread.h>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/mimo_hip.h"

namespace {

constexpr double kLog2Pi = 1.8378770664093454835606594728112;
constexpr double kLog2 = 0.69314718055994530941723212145818;

// psi(x) for x > 0: recurrence up to x >= 10, then the asymptotic series (error < 1e-15)
double digamma(double x) {
  double r = 0.0;
  while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
  const double f = 1.0 / (x * x);
  const double t = f * (-1.0 / 12.0 + f * (1.0 / 120.0 + f * (-1.0 / 252.0 + f * (1.0 / 240.0 + f * (-1.0 / 132.0
                   + f * (691.0 / 32760.0 + f * (-1.0 / 12.0)))))));
  return r + std::log(x) - 0.5 / x + t;
}

// SPD inverse of FOUR matrices at once, one per SIMD lane (the matrices are 2..33 wide: too short for
// row-wise vectorisation to pay, but K of them are independent).  Upper Cholesky A = U'U (right-looking),
// X = U^-1 by row back-substitution, A^-1 = X X'.  A[l] (n x n row-major, upper triangle read) ->
// Ainv[l] (full symmetric), sl[l] = sum log diag U (= 1/2 logdet A), NaN if A[l] is not positive definite.
// work: 2 n n v4d.
typedef double v4d __attribute__((vector_size(32)));

inline double dot(const double* __restrict__ x, const double* __restrict__ y, int n) {
#pragma clang fp reassociate(on)
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += x[i] * y[i];
  return s;
}

void spd_inverse4(const double* const A[4], int n, double* const Ainv[4], double sl[4], v4d* work) {
  v4d* U = work;
  v4d* X = work + (size_t)n * n;
  for (int i = 0; i < n; ++i)
    for (int j = i; j < n; ++j) U[i * n + j] = v4d{A[0][i * n + j], A[1][i * n + j], A[2][i * n + j], A[3][i * n + j]};
  bool bad[4] = {false, false, false, false};
  for (int l = 0; l < 4; ++l) sl[l] = 0.0;
  for (int j = 0; j < n; ++j) {
    v4d d = U[j * n + j], piv;
    for (int l = 0; l < 4; ++l) {
      if (!(d[l] > 0.0)) { bad[l] = true; d[l] = 1.0; }
      piv[l] = std::sqrt(d[l]);
      sl[l] += std::log(piv[l]);
    }
    const v4d inv = 1.0 / piv;
    v4d* uj = U + (size_t)j * n;
    for (int c = j; c < n; ++c) uj[c] *= inv;
    for (int i = j + 1; i < n; ++i) {
      const v4d f = uj[i];
      v4d* ui = U + (size_t)i * n;
      for (int c = i; c < n; ++c) ui[c] -= f * uj[c];
    }
  }
  const v4d zero = {0.0, 0.0, 0.0, 0.0}, one = {1.0, 1.0, 1.0, 1.0};
  for (int i = n - 1; i >= 0; --i) {           // row i of X = (e_i - sum_{k>i} U_ik X_k) / U_ii
    v4d* xi = X + (size_t)i * n;
    for (int c = i; c < n; ++c) xi[c] = zero;
    xi[i] = one;
    for (int k = i + 1; k < n; ++k) {
      const v4d f = U[i * n + k];
      const v4d* xk = X + (size_t)k * n;
      for (int c = k; c < n; ++c) xi[c] -= f * xk[c];
    }
    const v4d inv = 1.0 / U[i * n + i];
    for (int c = i; c < n; ++c) xi[c] *= inv;
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      v4d v = zero;
      const v4d* xi = X + (size_t)i * n;
      const v4d* xj = X + (size_t)j * n;
      for (int c = i; c < n; ++c) v += xi[c] * xj[c];
      for (int l = 0; l < 4; ++l) {
        Ainv[l][i * n + j] = v[l];
        Ainv[l][j * n + i] = v[l];
      }
    }
  for (int l = 0; l < 4; ++l)
    if (bad[l]) sl[l] = std::nan("");
}

double expected_logdet(double nu, int D, double half_logdet_psi) {   // wishart.py:139-143
  double s = 0.0;
  for (int i = 0; i < D; ++i) s += digamma(0.5 * (nu - i));
  return s + D * kLog2 + 2.0 * half_logdet_psi;
}

// Guard of the host-only entry points (no context, no message buffer): see mimo_abi.cpp.
template <typename F>
int guarded_host(F&& f) noexcept {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    return MIMO_E_NOMEM;
  } catch (...) {
    return MIMO_E_INTERNAL;
  }
}

std::atomic<int> g_fail_thread_start{0};   // mimo_host_debug_fault(4): the next helper-thread start throws

// ---- helper threads of the batched routines --------------------------------------------------------------------------------
// The K small dense problems of a sweep are split over threads when there is enough of them (K = 128, D = 32: 0.72 ms on one thread,
// 0.37 ms on eight).  The helpers persist: a pool created on first use (at most 7), asleep on a condition variable between calls; a call
// wakes what it needs with one notify, hands the groups out through one counter and waits for the helpers that took part (creating
// std::threads per call cost the CALLER ~15 us each).  The pool is a leaked singleton (threads
// blocked in it die with the process); after fork() the child starts its own on first use (pthread_atfork: the parent's threads do not
// exist there); a helper that cannot be created, or cannot allocate its scratch, leaves its share to the others.
class HelperPool {
 public:
  static HelperPool* get(int want) {
    std::lock_guard<std::mutex> lk(create_mu());
    HelperPool*& p = instance();
    if (!p) {
      static std::once_flag once;
      std::call_once(once, [] { pthread_atfork(nullptr, nullptr, [] { instance() = nullptr; new (&create_mu()) std::mutex(); }); });
      p = new (std::nothrow) HelperPool();
    }
    if (p) p->grow(want);
    return p;
  }
  int size() const { return size_.load(); }
  // runs job() on the caller and on up to `helpers` pool threads (<= size()); returns when every thread that took it is done.
  // job() returns when there is nothing left to hand out, so helpers that have not woken by the time the caller's own share is
  // finished are left asleep.  A second caller (another Python thread inside another routine) finds the pool busy and runs alone.
  void run(int helpers, const std::function<void()>& job) {
    std::unique_lock<std::mutex> call(call_mu_, std::try_to_lock);
    if (!call.owns_lock()) { job(); return; }
    {
      std::lock_guard<std::mutex> lk(mu_);
      job_ = &job; wanted_ = helpers; taken_ = 0; finished_ = 0; ++generation_;
    }
    cv_.notify_all();
    job();
    std::unique_lock<std::mutex> lk(mu_);
    wanted_ = taken_;                                      // late wakers find nothing to take
    done_cv_.wait(lk, [&] { return finished_ == taken_; });
    job_ = nullptr;
  }

 private:
  static HelperPool*& instance() { static HelperPool* p = nullptr; return p; }
  static std::mutex& create_mu() { static std::mutex* m = new std::mutex(); return *m; }
  void grow(int want) {
    want = std::min(want, 7);
    while ((int)threads_.size() < want) {
      try {
        threads_.emplace_back([this] { loop(); });
        threads_.back().detach();
        size_.store((int)threads_.size());
      } catch (...) { break; }                             // fewer helpers than wanted: the callers cope
    }
  }
  void loop() {
    int seen = 0;
    for (;;) {
      const std::function<void()>* job = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return generation_ != seen && taken_ < wanted_; });
        seen = generation_;
        ++taken_;
        job = job_;
      }
      if (job) (*job)();
      {
        std::lock_guard<std::mutex> lk(mu_);
        ++finished_;
      }
      done_cv_.notify_one();
    }
  }
  std::vector<std::thread> threads_;       // (grown under create_mu() only)
  std::atomic<int> size_{0};
  std::mutex mu_, call_mu_;
  std::condition_variable cv_, done_cv_;
  const std::function<void()>* job_ = nullptr;
  int wanted_ = 0, taken_ = 0, finished_ = 0, generation_ = 0;
};

// body(k0, count, work, Cbuf) handles components k0 .. k0+count-1 (count <= 4) with per-thread scratch
// (work: 2 n n v4d, Cbuf: 4 n n doubles) and returns false on a non-SPD block.
// The groups are handed out through one counter, so the call completes on however many threads take part: a helper that is
// missing (pool smaller than asked, mimo_host_debug_fault(4)) or cannot allocate its scratch leaves its share to the others.
// Groups are independent, so the result does not depend on which thread ran which group.
template <typename F>
int for_component_groups(int K, int n, double work_per_component, F&& body) {
  const int G = (K + 3) / 4;
  int nt = 1;   // ~1.4 ns per (4-lane) multiply-add; waking a helper costs 30 - 50 us until it runs (measured: K = 64, D = 16, 90 us of work,
                // ran 89 / 96 / 121 / 165 us on 1 / 2 / 4 / 8 threads): a helper per ~100 us of work = 7e4 multiply-adds, at most 8 threads
  {
    const unsigned hc = std::thread::hardware_concurrency();
    static const double per_thread = [] { const char* e = getenv("MIMO_HOST_WORK_PER_THREAD"); return e ? atof(e) : 7e4; }();   // tuning knobs
    static const int max_threads = [] { const char* e = getenv("MIMO_HOST_MAX_THREADS"); return e ? atoi(e) : 8; }();
    const double want = work_per_component * G / per_thread;
    nt = (int)std::min<double>(std::min<double>(hc ? hc : 1, max_threads), std::min<double>(G, want));
    if (nt < 1) nt = 1;
    if (g_fail_thread_start.load() && nt < 2 && G >= 2) nt = 2;   // the test hook wants a helper to be missing
  }
  std::atomic<int> next{0}, bad{0}, processed{0};
  auto run = [&]() noexcept {
    try {
      std::vector<v4d> work((size_t)2 * n * n);
      std::vector<double> Cbuf((size_t)4 * n * n);
      for (int g; (g = next.fetch_add(1)) < G;) {
        if (!body(4 * g, std::min(4, K - 4 * g), work.data(), Cbuf.data())) bad.store(1);
        processed.fetch_add(1);
      }
    } catch (...) {
      // scratch allocation failed on this thread: the others take its groups (none processed here)
    }
  };
  int helpers = nt - 1;
  if (helpers > 0 && g_fail_thread_start.exchange(0)) helpers = 0;      // (test hook: the helpers "could not be started")
  HelperPool* pool = helpers > 0 ? HelperPool::get(helpers) : nullptr;
  if (pool) helpers = std::min(helpers, pool->size());
  if (pool && helpers > 0) {
    const std::function<void()> job = run;
    pool->run(helpers, job);
  } else {
    run();
  }
  if (processed.load() != G) return MIMO_E_NOMEM;   // no thread could allocate its scratch
  return bad.load() ? MIMO_E_INVALID : MIMO_OK;
}

}  // namespace

// ---- numpy.random's legacy stream (RandomState over MT19937), restated -----------------------------------------------
// What numpy/random/src/mt19937/mt19937.c and src/legacy/legacy-distributions.c compute, so that the per-component draws the
// reference makes (wishart.py:72-92: normal(n) then chisquare(nu - i) per i; gamma.py:53-55) can be taken K blocks at a time in
// native code and still leave numpy.random's generator exactly where K x (2 + D) Python calls would have left it.  The value of
// every variate depends on the order and rounding of a handful of float64 operations: no contraction to fused multiply-adds here
// (numpy's baseline build has none), libm's log / sqrt / pow as numpy calls them.
#if defined(__clang__)
#define MIMO_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define MIMO_NO_CONTRACT
#endif
#if defined(__GNUC__) && !defined(__clang__)
#define MIMO_NO_CONTRACT_FN __attribute__((optimize("fp-contract=off")))
#else
#define MIMO_NO_CONTRACT_FN
#endif

struct LegacyStream {
  uint32_t* key;      // [624]
  int pos;
  int has_gauss;
  double gauss;
  // where the generator stood when the pair behind the cached gaussian was begun (mimo_host_legacy_draws_inplace hands the cached
  // value back to numpy by rewinding to that point and letting numpy draw the pair again): position, refills seen so far, and the
  // key as it was before the latest refill (one 2.5 KB copy per 624 outputs)
  uint32_t* backup = nullptr;     // [624] or null: no bookkeeping
  int refills = 0, pair_pos = 0, pair_refills = 0, pairs = 0;

  void refill() {
    if (backup) { std::memcpy(backup, key, 624 * sizeof(uint32_t)); ++refills; }
    constexpr int N = 624, M = 397;
    constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MAG = 0x9908b0dfu;
    int kk = 0;
    uint32_t y;
    for (; kk < N - M; ++kk) {
      y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
      key[kk] = key[kk + M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MAG);
    }
    for (; kk < N - 1; ++kk) {
      y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
      key[kk] = key[kk + (M - N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MAG);
    }
    y = (key[N - 1] & UPPER) | (key[0] & LOWER);
    key[N - 1] = key[M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MAG);
    pos = 0;
  }
  uint32_t next32() {
    if (pos == 624) refill();
    uint32_t y = key[pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
  MIMO_NO_CONTRACT_FN double next_double() {
    MIMO_NO_CONTRACT
    const int32_t a = (int32_t)(next32() >> 5), b = (int32_t)(next32() >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
  }
  MIMO_NO_CONTRACT_FN double gauss_next() {            // legacy_gauss: polar method, the second variate kept for the next call
    MIMO_NO_CONTRACT
    if (has_gauss) {
      const double t = gauss;
      has_gauss = 0;
      gauss = 0.0;
      return t;
    }
    double f, x1, x2, r2;
    pair_pos = pos; pair_refills = refills; ++pairs;
    do {
      x1 = 2.0 * next_double() - 1.0;
      x2 = 2.0 * next_double() - 1.0;
      r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = std::sqrt(-2.0 * std::log(r2) / r2);
    gauss = f * x1;
    has_gauss = 1;
    return f * x2;
  }
  MIMO_NO_CONTRACT_FN double standard_exponential() {
    MIMO_NO_CONTRACT
    return -std::log(1.0 - next_double());
  }
  MIMO_NO_CONTRACT_FN double standard_gamma(double shape) {   // legacy_standard_gamma (Marsaglia & Tsang above 1)
    MIMO_NO_CONTRACT
    double b, c, U, V, X, Y;
    if (shape == 1.0) return standard_exponential();
    if (shape == 0.0) return 0.0;
    if (shape < 1.0) {
      for (;;) {
        U = next_double();
        V = standard_exponential();
        if (U <= 1.0 - shape) {
          X = std::pow(U, 1. / shape);
          if (X <= V) return X;
        } else {
          Y = -std::log((1 - U) / shape);
          X = std::pow(1.0 - shape + shape * Y, 1. / shape);
          if (X <= (V + Y)) return X;
        }
      }
    }
    b = shape - 1. / 3.;
    c = 1. / std::sqrt(9 * b);
    for (;;) {
      do {
        X = gauss_next();
        V = 1.0 + c * X;
      } while (V <= 0.0);
      V = V * V * V;
      U = next_double();
      if (U < 1.0 - 0.0331 * (X * X) * (X * X)) return (b * V);
      if (std::log(U) < 0.5 * X * X + b * (1. - V + std::log(V))) return (b * V);
    }
  }
};

extern "C" {

int mimo_host_nw_vi(int K, int D, const double* a, const double* b, const double* c, const double* d,
                    double* mus, double* psis, double* nus, double* half_logdet_psi,
                    double* cc, double* bb, double* W, double* E2, double* E4) {
  return guarded_host([&]() -> int {
  if (K < 1 || D < 1 || !a || !b || !c || !d || !mus || !psis || !nus || !half_logdet_psi || !cc || !bb || !W ||
      !E2 || !E4)
    return MIMO_E_INVALID;
  return for_component_groups(K, D, (double)D * D * D, [&](int k0, int cnt, v4d* work, double* Cbuf) {
    const double* Ain[4];
    double* Aout[4];
    double sl[4];
    for (int l = 0; l < 4; ++l) {
      const int k = k0 + std::min(l, cnt - 1);          // tail lanes repeat the last component
      const double kap = b[k];
      double* m = mus + (size_t)k * D;
      if (l < cnt)
        for (int i = 0; i < D; ++i) m[i] = a[(size_t)k * D + i] / kap;
      const double* ck = c + (size_t)k * D * D;
      double* C = Cbuf + (size_t)l * D * D;
      for (int i = 0; i < D; ++i)
        for (int j = i; j < D; ++j) C[i * D + j] = ck[i * D + j] - kap * m[i] * m[j];
      Ain[l] = C;
      Aout[l] = l < cnt ? psis + (size_t)k * D * D : C;   // spare lanes write into their own scratch
    }
    spd_inverse4(Ain, D, Aout, sl, work);
    bool ok = true;
    for (int l = 0; l < cnt; ++l) {
      const int k = k0 + l;
      if (std::isnan(sl[l])) { ok = false; continue; }
      const double kap = b[k];
      const double* m = mus + (size_t)k * D;
      const double* psi = psis + (size_t)k * D * D;
      const double nu = d[k] + D;
      nus[k] = nu;
      half_logdet_psi[k] = -sl[l];
      double* Wk = W + (size_t)k * D * D;
      double* bk = bb + (size_t)k * D;
      for (int i = 0; i < D * D; ++i) Wk[i] = nu * psi[i];
      double mWm = 0.0;
      for (int i = 0; i < D; ++i) {
        const double s = dot(Wk + (size_t)i * D, m, D);
        bk[i] = s;
        mWm += m[i] * s;
      }
      E2[k] = -0.5 * (D / kap + mWm);
      E4[k] = 0.5 * expected_logdet(nu, D, -sl[l]);
      cc[k] = -0.5 * D * kLog2Pi + E2[k] + E4[k];
    }
    return ok;
  });
  });
}

int mimo_host_nw_vi_tied(int K, int D, const double* a, const double* b, const double* c, const double* d,
                         double* mus, double* psis, double* nus, double* half_logdet_psi, double* nat_c,
                         double* cc, double* bb, double* W, double* E2, double* E4) {
  return guarded_host([&]() -> int {
  if (K < 1 || D < 1 || !a || !b || !c || !d || !mus || !psis || !nus || !half_logdet_psi || !nat_c || !cc ||
      !bb || !W || !E2 || !E4)
    return MIMO_E_INVALID;
  const size_t DD = (size_t)D * D;
  std::vector<double> P(DD, 0.0), psi(DD), Pinv(DD), scratch(3 * DD);
  std::vector<v4d> work(2 * DD);
  double nu = 0.0;
  for (int k = 0; k < K; ++k) {                      // the pooled block, summed over k in order (np.mean, axis 0)
    const double kap = b[k];
    double* m = mus + (size_t)k * D;
    for (int i = 0; i < D; ++i) m[i] = a[(size_t)k * D + i] / kap;
    const double* ck = c + (size_t)k * DD;
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) P[i * D + j] += ck[i * D + j] - kap * (m[i] * m[j]);
    nu += d[k] + D;
  }
  for (size_t i = 0; i < DD; ++i) P[i] /= K;
  nu /= K;
  double sl[4];
  {                                                   // psi = P^-1, then psi^-1 as std_to_nat reads it back
    double* spare[3] = {scratch.data(), scratch.data() + DD, scratch.data() + 2 * DD};   // the three idle lanes
    const double* Ain[4] = {P.data(), P.data(), P.data(), P.data()};
    double* Aout[4] = {psi.data(), spare[0], spare[1], spare[2]};
    spd_inverse4(Ain, D, Aout, sl, work.data());
    if (std::isnan(sl[0])) return MIMO_E_INVALID;
    const double hld = -sl[0];
    const double* Bin[4] = {psi.data(), psi.data(), psi.data(), psi.data()};
    double* Bout[4] = {Pinv.data(), spare[0], spare[1], spare[2]};
    spd_inverse4(Bin, D, Bout, sl, work.data());
    if (std::isnan(sl[0])) return MIMO_E_INVALID;
    const double e4 = 0.5 * expected_logdet(nu, D, hld);
    for (int k = 0; k < K; ++k) {
      const double kap = b[k];
      const double* m = mus + (size_t)k * D;
      double* Wk = W + (size_t)k * DD;
      double* bk = bb + (size_t)k * D;
      double* nk = nat_c + (size_t)k * DD;
      std::copy(psi.begin(), psi.end(), psis + (size_t)k * DD);
      nus[k] = nu;
      half_logdet_psi[k] = hld;
      for (size_t i = 0; i < DD; ++i) Wk[i] = nu * psi[i];
      for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) nk[i * D + j] = Pinv[i * D + j] + kap * (m[i] * m[j]);
      double mWm = 0.0;
      for (int i = 0; i < D; ++i) {
        const double s = dot(Wk + (size_t)i * D, m, D);
        bk[i] = s;
        mWm += m[i] * s;
      }
      E2[k] = -0.5 * (D / kap + mWm);
      E4[k] = e4;
      cc[k] = -0.5 * D * kLog2Pi + E2[k] + E4[k];
    }
  }
  return MIMO_OK;
  });
}

int mimo_host_nw_vlb(int K, int D, const double* qa, const double* qb, const double* qc, const double* qd,
                     const double* pa, const double* pb, const double* pc, const double* pd, const double* prior_logZ,
                     const double* nus, const double* half_logdet_psi, const double* E1, const double* E2,
                     const double* W, const double* E4, double* out) {
  return guarded_host([&]() -> int {
  if (K < 1 || D < 1 || !qa || !qb || !qc || !qd || !pa || !pb || !pc || !pd || !prior_logZ || !nus ||
      !half_logdet_psi || !E1 || !E2 || !W || !E4 || !out)
    return MIMO_E_INVALID;
  const size_t DD = (size_t)D * D;
  const double log_base = -0.5 * D * kLog2Pi;
  for (int k = 0; k < K; ++k) {
    const double nu = nus[k];
    double mg = 0.25 * D * (D - 1) * std::log(M_PI);          // ln Gamma_D(nu / 2)
    for (int j = 0; j < D; ++j) { int sign; mg += lgamma_r(0.5 * nu - 0.5 * j, &sign); }
    const double logZq = -0.5 * D * std::log(qb[k]) + (0.5 * nu * D * kLog2 + mg + nu * half_logdet_psi[k]);
    const double* e1 = E1 + (size_t)k * D;
    const double* w = W + (size_t)k * DD;
    // <eta, E[t]> with E3 = -W/2 (composite.py:120-128), for the posterior's and the prior's eta
    const double iq = dot(qa + (size_t)k * D, e1, D) + qb[k] * E2[k] - 0.5 * dot(qc + (size_t)k * DD, w, (int)DD) + qd[k] * E4[k];
    const double ip = dot(pa + (size_t)k * D, e1, D) + pb[k] * E2[k] - 0.5 * dot(pc + (size_t)k * DD, w, (int)DD) + pd[k] * E4[k];
    out[k] = (logZq - log_base - iq) - (prior_logZ[k] - log_base - ip);
  }
  return MIMO_OK;
  });
}

int mimo_host_gmm_vi_sweep(int K, int D, int tied, const double* alpha0, const double* counts,
                           const double* pa, const double* pb, const double* pc, const double* pd,
                           const double* sx, const double* sn, const double* sxx,
                           double* alpha, double* qa, double* qb, double* qc, double* qd,
                           double* mus, double* psis, double* nus, double* half_logdet_psi, double* nat_c,
                           double* cc, double* bb, double* W, double* E2, double* E4, double* e_log_pi, double* c_total) {
  return guarded_host([&]() -> int {
  if (K < 1 || D < 1 || !alpha0 || !counts || !pa || !pb || !pc || !pd || !sx || !sn || !sxx || !alpha || !qa || !qb ||
      !qc || !qd || !mus || !psis || !nus || !half_logdet_psi || (tied && !nat_c) || !cc || !bb || !W || !E2 || !E4 ||
      !e_log_pi || !c_total)
    return MIMO_E_INVALID;
  const size_t DD = (size_t)D * D;
  for (size_t i = 0; i < (size_t)K * D; ++i) qa[i] = pa[i] + sx[i];           // eta_post = eta_prior + statistics
  for (int k = 0; k < K; ++k) { qb[k] = pb[k] + sn[k]; qd[k] = pd[k] + sn[k]; }
  for (size_t i = 0; i < (size_t)K * DD; ++i) qc[i] = pc[i] + sxx[i];
  const int rc = tied ? mimo_host_nw_vi_tied(K, D, qa, qb, qc, qd, mus, psis, nus, half_logdet_psi, nat_c, cc, bb, W, E2, E4)
                      : mimo_host_nw_vi(K, D, qa, qb, qc, qd, mus, psis, nus, half_logdet_psi, cc, bb, W, E2, E4);
  if (rc != MIMO_OK) return rc;
  // Dirichlet gating (dirichlet.py:31-33,85-87; bayesian.py:78-83): alpha = ((alpha0 - 1) + counts) + 1
  double s = 0.0;
  for (int k = 0; k < K; ++k) { alpha[k] = ((alpha0[k] - 1.0) + counts[k]) + 1.0; s += alpha[k]; }
  const double ds = digamma(s);
  for (int k = 0; k < K; ++k) {
    e_log_pi[k] = digamma(alpha[k]) - ds;
    c_total[k] = cc[k] + e_log_pi[k];
  }
  return MIMO_OK;
  });
}

int mimo_host_gmm_vi_bound(int K, int D, int tied, const double* alpha0, const double* alpha, const double* e_log_pi,
                           const double* pa, const double* pb, const double* pc, const double* pd, const double* prior_logZ,
                           const double* qa, const double* qb, const double* qc, const double* qd,
                           const double* mus, const double* nus, const double* half_logdet_psi, const double* nat_c,
                           const double* bb, const double* E2, const double* W, const double* E4, double* vlb) {
  return guarded_host([&]() -> int {
  if (K < 1 || D < 1 || !alpha0 || !alpha || !e_log_pi || !pa || !pb || !pc || !pd || !prior_logZ || !qa || !qb || !qc ||
      !qd || !mus || !nus || !half_logdet_psi || (tied && !nat_c) || !bb || !E2 || !W || !E4 || !vlb)
    return MIMO_E_INVALID;
  std::vector<double> per_k(K), ta, td;
  const double *va = qa, *vc = qc, *vd = qd;
  if (tied) {                       // the tied block's natural parameters are read back from the pooled standard ones
    ta.resize((size_t)K * D); td.resize(K);
    for (int k = 0; k < K; ++k) {
      for (int i = 0; i < D; ++i) ta[(size_t)k * D + i] = qb[k] * mus[(size_t)k * D + i];
      td[k] = nus[k] - D;
    }
    va = ta.data(); vc = nat_c; vd = td.data();
  }
  const int rc = mimo_host_nw_vlb(K, D, va, qb, vc, vd, pa, pb, pc, pd, prior_logZ, nus, half_logdet_psi, bb, E2, W, E4,
                                  per_k.data());
  if (rc != MIMO_OK) return rc;
  double comp = 0.0;
  for (int k = 0; k < K; ++k) comp += per_k[k];
  double s = 0.0, s0 = 0.0, lq = 0.0, lp = 0.0, iq = 0.0, ip = 0.0;       // dirichlet.py:78-79,89-97
  int sign;
  for (int k = 0; k < K; ++k) {
    s += alpha[k]; s0 += alpha0[k];
    lq += lgamma_r(alpha[k], &sign); lp += lgamma_r(alpha0[k], &sign);
    iq += (alpha[k] - 1.0) * e_log_pi[k];
    ip += (alpha0[k] - 1.0) * e_log_pi[k];
  }
  vlb[0] = ((lq - lgamma_r(s, &sign)) - iq) - ((lp - lgamma_r(s0, &sign)) - ip);
  vlb[1] = comp;
  return MIMO_OK;
  });
}

MIMO_NO_CONTRACT_FN int mimo_host_legacy_draws(uint32_t* mt_key, int* mt_pos, int* has_gauss, double* gauss, int K, int n_before,
                                               int n_gamma, int n_after, const double* shapes, double* before, double* gammas,
                                               double* after) {
  return guarded_host([&]() -> int {
  MIMO_NO_CONTRACT
  if (!mt_key || !mt_pos || !has_gauss || !gauss || K < 0 || n_before < 0 || n_gamma < 0 || n_after < 0 ||
      *mt_pos < 0 || *mt_pos > 624 || (n_gamma && !shapes) || (n_before && !before) || (n_gamma && !gammas) || (n_after && !after))
    return MIMO_E_INVALID;
  for (size_t i = 0; i < (size_t)K * n_gamma; ++i)
    if (!(shapes[i] >= 0.0)) return MIMO_E_INVALID;       // (numpy raises on a negative or NaN shape: the caller's route)
  LegacyStream g{mt_key, *mt_pos, *has_gauss, *gauss};
  for (int k = 0; k < K; ++k) {
    for (int i = 0; i < n_before; ++i) before[(size_t)k * n_before + i] = 0.0 + 1.0 * g.gauss_next();   // legacy_normal(0, 1)
    for (int i = 0; i < n_gamma; ++i) gammas[(size_t)k * n_gamma + i] = g.standard_gamma(shapes[(size_t)k * n_gamma + i]);
    for (int i = 0; i < n_after; ++i) after[(size_t)k * n_after + i] = 0.0 + 1.0 * g.gauss_next();
  }
  *mt_pos = g.pos; *has_gauss = g.has_gauss; *gauss = g.gauss;
  return MIMO_OK;
  });
}

int mimo_host_hier_vi(int K, int D, int nb_iter, const double* kap, const double* m0, double kappa0, const double* psi0_inv,
                      double nu0, const double* xk, const double* nk, const double* sxx_sum, double* mu_q, double* post_mus,
                      double* post_kappas, double* kappa_q, double* psi_q, double* nu_q) {
  return guarded_host([&]() -> int {
  if (K < 1 || D < 1 || nb_iter < 1 || !kap || !m0 || !psi0_inv || !xk || !nk || !sxx_sum || !mu_q || !post_mus || !post_kappas ||
      !kappa_q || !psi_q || !nu_q)
    return MIMO_E_INVALID;
  const size_t DD = (size_t)D * D;
  std::vector<double> pooled(DD), A(DD), scratch(3 * DD), wsh(K), num(D);
  std::vector<v4d> work(2 * DD);
  double ksum = 0.0, nusum = 0.0;
  for (int k = 0; k < K; ++k) {
    post_kappas[k] = kap[k] + nk[k];
    wsh[k] = kappa0 * kap[k] / (kappa0 + kap[k]);
    ksum += kap[k] + kappa0;
    nusum += nu0 + nk[k] + 1.0;
  }
  for (size_t i = 0; i < DD; ++i) pooled[i] = psi0_inv[i] + sxx_sum[i] / K;
  for (int it = 0; it < nb_iter; ++it) {
    // component means given the hyper-posterior mean, then the pooled hyper-posterior given the means (bayesian.py:661-689)
    for (int k = 0; k < K; ++k)
      for (int i = 0; i < D; ++i)
        post_mus[(size_t)k * D + i] = (kap[k] * mu_q[i] + xk[(size_t)k * D + i]) / post_kappas[k];
    for (int i = 0; i < D; ++i) num[i] = 0.0;
    for (size_t i = 0; i < DD; ++i) A[i] = 0.0;
    std::vector<double> dv(D);
    for (int k = 0; k < K; ++k) {
      const double* __restrict__ m = post_mus + (size_t)k * D;
      const double* __restrict__ x = xk + (size_t)k * D;
      double* __restrict__ d = dv.data();
      for (int i = 0; i < D; ++i) { num[i] += kap[k] * m[i] + kappa0 * m0[i]; d[i] = m0[i] - m[i]; }
      for (int i = 0; i < D; ++i) {
        const double wd = wsh[k] * d[i], mi = m[i], xi = x[i], nm = nk[k] * m[i];
        double* __restrict__ Ai = A.data() + (size_t)i * D;
        for (int j = i; j < D; ++j)      // upper triangle of shrink - cross - cross' + sum_k n_k m m'
          Ai[j] += wd * d[j] - mi * x[j] - xi * m[j] + nm * m[j];
      }
    }
    for (int i = 0; i < D; ++i) mu_q[i] = num[i] / ksum;
    for (int i = 0; i < D; ++i)
      for (int j = i; j < D; ++j) A[i * D + j] = pooled[i * D + j] + A[i * D + j] / K;
    const double* Ain[4] = {A.data(), A.data(), A.data(), A.data()};
    double* Aout[4] = {psi_q, scratch.data(), scratch.data() + DD, scratch.data() + 2 * DD};
    double sl[4];
    spd_inverse4(Ain, D, Aout, sl, work.data());
    if (std::isnan(sl[0])) return MIMO_E_INVALID;
  }
  *kappa_q = ksum / K;
  *nu_q = nusum / K;
  return MIMO_OK;
  });
}

MIMO_NO_CONTRACT_FN int mimo_host_legacy_draws_inplace(uint32_t* mt_key, int* mt_pos, int has_gauss, double gauss, int K, int n_before,
                                                       int n_gamma, int n_after, const double* shapes, double* before, double* gammas,
                                                       double* after, int* redraw, uint32_t* final_state) {
  return guarded_host([&]() -> int {
  MIMO_NO_CONTRACT
  if (!mt_key || !mt_pos || !redraw || !final_state || K < 0 || n_before < 0 || n_gamma < 0 || n_after < 0 || *mt_pos < 0 || *mt_pos > 624 ||
      (n_gamma && !shapes) || (n_before && !before) || (n_gamma && !gammas) || (n_after && !after))
    return MIMO_E_INVALID;
  for (size_t i = 0; i < (size_t)K * n_gamma; ++i)
    if (!(shapes[i] >= 0.0)) return MIMO_E_INVALID;
  std::vector<uint32_t> backup(624);
  LegacyStream g{mt_key, *mt_pos, has_gauss, gauss};
  g.backup = backup.data();
  for (int k = 0; k < K; ++k) {
    for (int i = 0; i < n_before; ++i) before[(size_t)k * n_before + i] = 0.0 + 1.0 * g.gauss_next();
    for (int i = 0; i < n_gamma; ++i) gammas[(size_t)k * n_gamma + i] = g.standard_gamma(shapes[(size_t)k * n_gamma + i]);
    for (int i = 0; i < n_after; ++i) after[(size_t)k * n_after + i] = 0.0 + 1.0 * g.gauss_next();
  }
  *redraw = 0;
  if (g.has_gauss) {
    // a gaussian is left cached: numpy must hold it.  Rewind to where its pair was begun; the caller draws one normal from numpy,
    // which produces the pair again, returns the half already used here and caches this one.  (No pair begun here: the cached value
    // is the caller's own, untouched — a request without a single gaussian, which the caller keeps away from this entry point.)
    if (g.pairs == 0) return MIMO_E_STATE;
    // (uniforms may have been drawn after that pair: the state reached is handed out in final_state — key[624], pos — for the caller
    // to write back once numpy holds the gaussian)
    std::memcpy(final_state, mt_key, 624 * sizeof(uint32_t));
    final_state[624] = (uint32_t)g.pos;
    if (g.refills > g.pair_refills) std::memcpy(mt_key, backup.data(), 624 * sizeof(uint32_t));   // (at most one refill since the pair began:
    *mt_pos = g.pair_pos;                                                                          //  a refill yields 624 outputs)
    *redraw = 1;
    return MIMO_OK;
  }
  *mt_pos = g.pos;
  return MIMO_OK;
  });
}

int mimo_host_py_sample(uint32_t* mt_key, int* mt_pos, int64_t n, int64_t k, int use_pool, int64_t* out) {
  return guarded_host([&]() -> int {
  if (!mt_key || !mt_pos || !out || k < 0 || k > n || n < 1 || n >= ((int64_t)1 << 32) || *mt_pos < 0 || *mt_pos > 624) return MIMO_E_INVALID;
  LegacyStream g{mt_key, *mt_pos, 0, 0.0};
  auto randbelow = [&](uint64_t m) -> uint64_t {        // Random._randbelow_with_getrandbits (m >= 1): getrandbits(m.bit_length()) until < m
    int bits = 0;
    for (uint64_t t = m; t; t >>= 1) ++bits;
    uint64_t r;
    do { r = (uint64_t)(g.next32() >> (32 - bits)); } while (r >= m);
    return r;
  };
  if (use_pool) {                                       // "an n-length list is smaller than a k-length set"
    std::vector<int64_t> pool((size_t)n);
    for (int64_t i = 0; i < n; ++i) pool[(size_t)i] = i;
    for (int64_t i = 0; i < k; ++i) {
      const uint64_t j = randbelow((uint64_t)(n - i));
      out[i] = pool[(size_t)j];
      pool[(size_t)j] = pool[(size_t)(n - i - 1)];      // move the non-selected item into the vacancy
    }
  } else {                                              // selections tracked in a set: open addressing, load <= 1/2
    size_t cap = 16;
    while (cap < (size_t)(2 * k + 1)) cap <<= 1;
    std::vector<int64_t> table(cap, -1);
    auto insert = [&](int64_t v) -> bool {              // false: already there
      size_t h = (size_t)((uint64_t)v * 0x9E3779B97F4A7C15ull) & (cap - 1);
      while (table[h] >= 0) { if (table[h] == v) return false; h = (h + 1) & (cap - 1); }
      table[h] = v;
      return true;
    };
    for (int64_t i = 0; i < k; ++i) {
      int64_t j = (int64_t)randbelow((uint64_t)n);
      while (!insert(j)) j = (int64_t)randbelow((uint64_t)n);
      out[i] = j;
    }
  }
  *mt_pos = g.pos;
  return MIMO_OK;
  });
}

int mimo_host_mnw_vi(int K, int dy, int dc, int affine, const double* a, const double* b, const double* c,
                     const double* d, double* Ms, double* psis, double* nus, double* half_logdet_psi,
                     double* Kinv, double* cc, double* bb, double* W, double* E1, double* E2, double* E4) {
  return guarded_host([&]() -> int {
  if (K < 1 || dy < 1 || dc < 1 || (affine && dc < 2) || !a || !b || !c || !d || !Ms || !psis || !nus ||
      !half_logdet_psi || !Kinv || !cc || !bb || !W || !E1 || !E2 || !E4)
    return MIMO_E_INVALID;
  const int dx = affine ? dc - 1 : dc, Dz = dx + dy;
  const int n = dc > dy ? dc : dy;
  return for_component_groups(K, n, (double)n * n * n * 3, [&](int k0, int cnt, v4d* work, double* Cbuf) {
    const double* Ain[4];
    double* Aout[4];
    double sl[4];
    for (int l = 0; l < 4; ++l) {                    // K_k^-1
      const int k = k0 + std::min(l, cnt - 1);
      Ain[l] = b + (size_t)k * dc * dc;
      Aout[l] = l < cnt ? Kinv + (size_t)k * dc * dc : Cbuf + (size_t)l * n * n;
    }
    spd_inverse4(Ain, dc, Aout, sl, work);
    bool ok = true;
    for (int l = 0; l < cnt; ++l) ok = ok && !std::isnan(sl[l]);
    if (!ok) return false;
    for (int l = 0; l < 4; ++l) {                    // M = a K^-1, psi^-1 = c - a M'
      const int k = k0 + std::min(l, cnt - 1);
      const double* ak = a + (size_t)k * dy * dc;
      const double* Ki = Kinv + (size_t)k * dc * dc;
      double* M = Ms + (size_t)k * dy * dc;
      double* C = Cbuf + (size_t)l * n * n;
      if (l < cnt)
        for (int i = 0; i < dy; ++i)
          for (int j = 0; j < dc; ++j) {
            double s = 0.0;
            for (int q = 0; q < dc; ++q) s += ak[i * dc + q] * Ki[q * dc + j];
            M[i * dc + j] = s;
          }
      const double* ck = c + (size_t)k * dy * dy;
      for (int i = 0; i < dy; ++i)
        for (int j = i; j < dy; ++j) {
          double s = ck[i * dy + j];
          for (int q = 0; q < dc; ++q) s -= ak[i * dc + q] * M[j * dc + q];
          C[i * dy + j] = s;
        }
      Ain[l] = C;
      Aout[l] = l < cnt ? psis + (size_t)k * dy * dy : C;
    }
    spd_inverse4(Ain, dy, Aout, sl, work);
    for (int l = 0; l < cnt; ++l) {
      const int k = k0 + l;
      if (std::isnan(sl[l])) { ok = false; continue; }
      const double* Ki = Kinv + (size_t)k * dc * dc;
      const double* M = Ms + (size_t)k * dy * dc;
      const double* psi = psis + (size_t)k * dy * dy;
      const double nu = d[k] + dy + 1.0 - dc;
      nus[k] = nu;
      half_logdet_psi[k] = -sl[l];
      // expected statistics (composite.py:635-647)
      double* e1 = E1 + (size_t)k * dy * dc;
      double* e2 = E2 + (size_t)k * dc * dc;
      for (int i = 0; i < dy; ++i)
        for (int j = 0; j < dc; ++j) {
          double s = 0.0;
          for (int q = 0; q < dy; ++q) s += psi[i * dy + q] * M[q * dc + j];
          e1[i * dc + j] = nu * s;
        }
      for (int i = 0; i < dc; ++i)
        for (int j = 0; j < dc; ++j) {
          double s = 0.0;
          for (int q = 0; q < dy; ++q) s += M[q * dc + i] * e1[q * dc + j];
          e2[i * dc + j] = -0.5 * (dy * Ki[i * dc + j] + s);
        }
      E4[k] = 0.5 * expected_logdet(nu, dy, -sl[l]);
      // canonical (c, b, W) over z = [x, y]  (bayesian.py:933-947)
      double* Wk = W + (size_t)k * Dz * Dz;
      double* bz = bb + (size_t)k * Dz;
      for (int i = 0; i < dx; ++i)
        for (int j = 0; j < dx; ++j) Wk[i * Dz + j] = -2.0 * e2[i * dc + j];
      for (int i = 0; i < dy; ++i)
        for (int j = 0; j < dy; ++j) Wk[(dx + i) * Dz + dx + j] = nu * psi[i * dy + j];
      for (int i = 0; i < dy; ++i)
        for (int j = 0; j < dx; ++j) {
          Wk[(dx + i) * Dz + j] = -e1[i * dc + j];
          Wk[j * Dz + dx + i] = -e1[i * dc + j];
        }
      double c0 = -0.5 * dy * kLog2Pi + E4[k];
      if (affine) {
        for (int j = 0; j < dx; ++j) bz[j] = 2.0 * e2[j * dc + dx];
        for (int i = 0; i < dy; ++i) bz[dx + i] = e1[i * dc + dx];
        c0 += e2[dx * dc + dx];
      } else {
        for (int j = 0; j < Dz; ++j) bz[j] = 0.0;
      }
      cc[k] = c0;
    }
    return ok;
  });
  });
}

int mimo_host_nw_gibbs(int K, int D, const double* mus, const double* kappas, const double* psis,
                       const double* z, const double* g, const double* eps,
                       double* out_mu, double* out_lmbda, double* out_c, double* out_b) {
  return guarded_host([&]() -> int {
  if (K < 1 || D < 1 || D > 64) return MIMO_E_INVALID;
  const int nt = D * (D - 1) / 2;
  std::vector<double> Lb((size_t)D * D), Tb((size_t)D * D);
  double* L = Lb.data();
  double* T = Tb.data();
  double y[64];
  for (int k = 0; k < K; ++k) {
    const double* psi = psis + (size_t)k * D * D;
    // psi = L L' (lower; the upper triangle of psi is not read)
    for (int i = 0; i < D; ++i)
      for (int j = 0; j <= i; ++j) {
        double s = psi[i * D + j];
        for (int m = 0; m < j; ++m) s -= L[i * D + m] * L[j * D + m];
        if (i == j) {
          if (!(s > 0.0)) return MIMO_E_INVALID;
          L[i * D + i] = std::sqrt(s);
        } else {
          L[i * D + j] = s / L[j * D + j];
        }
      }
    // Bartlett factor A (lower: sqrt-chi-square diagonal, standard normals below it, in the order of
    // numpy.tril_indices(D, -1)) and T = L A: Lambda = T T', and T IS the lower Cholesky factor of Lambda
    const double* zk = z + (size_t)k * nt;
    const double* gk = g + (size_t)k * D;
    for (int i = 0; i < D; ++i)
      for (int j = 0; j <= i; ++j) {
        double s = L[i * D + j] * gk[j];                                   // m = j: A_jj
        for (int m = j + 1; m <= i; ++m) s += L[i * D + m] * zk[m * (m - 1) / 2 + j];
        T[i * D + j] = s;
      }
    double* lam = out_lmbda + (size_t)k * D * D;
    for (int i = 0; i < D; ++i)
      for (int j = 0; j <= i; ++j) {
        double s = 0.0;
        for (int m = 0; m <= j; ++m) s += T[i * D + m] * T[j * D + m];
        lam[i * D + j] = s;
        lam[j * D + i] = s;
      }
    // mu = m + (sqrt(kappa) T)^-T eps  (covariance (kappa Lambda)^-1): back substitution on T'
    const double rk = 1.0 / std::sqrt(kappas[k]);
    const double* ek = eps + (size_t)k * D;
    double sld = 0.0;
    for (int i = D - 1; i >= 0; --i) {
      double s = ek[i] * rk;
      for (int m = i + 1; m < D; ++m) s -= T[m * D + i] * y[m];
      y[i] = s / T[i * D + i];
      sld += std::log(T[i * D + i]);
    }
    double* mu = out_mu + (size_t)k * D;
    double* bk = out_b + (size_t)k * D;
    for (int i = 0; i < D; ++i) mu[i] = mus[(size_t)k * D + i] + y[i];
    double quad = 0.0;
    for (int i = 0; i < D; ++i) {
      double s = 0.0;
      for (int j = 0; j < D; ++j) s += lam[i * D + j] * mu[j];
      bk[i] = s;
      quad += s * mu[i];
    }
    out_c[k] = -0.5 * quad + sld - 0.5 * D * kLog2Pi;
  }
  return MIMO_OK;
  });
}

// Position-dependent content checksum of a byte range, the same function the device computes over the resident rows
// (nan_any_kernel in mimo_small.hip: mimo_data_checksum).  Words w_i = the 8-byte little-endian words (the last one
// zero-extended), m_i = w_i ^ (w_i >> 32) (a bijection that carries the high half — where "round" doubles such as 0.0, 1.0, 2.0
// differ — into the low bits), nw = their number:
//     out[0] = sum_i m_i              out[1] = sum_i (nw - i) m_i          (mod 2^64)
// Any edit of one word changes both; a swap of two unequal words changes out[1] (their distance is below 2^32 words); sums
// mod 2^64 do not depend on how the range is cut into slices or in which order the slices are added.
int mimo_host_checksum(const void* data, size_t nbytes, uint64_t out[2]) {
  return guarded_host([&]() -> int {
  if ((!data && nbytes) || !out) return MIMO_E_INVALID;
  const unsigned char* p = static_cast<const unsigned char*>(data);
  const size_t nfull = nbytes / 8, nw = (nbytes + 7) / 8;
  // slices of whole words; a thread per ~4 MB, at most 8 (a thread costs ~15 us to start: 1 MB of streaming)
  int nt = (int)std::min<size_t>(8, nbytes / ((size_t)4 << 20));
  {
    const unsigned hc = std::thread::hardware_concurrency();
    nt = std::max(1, std::min<int>(nt, hc ? (int)hc : 1));
  }
  std::vector<uint64_t> part((size_t)2 * nt, 0);
  auto run = [&](int t) noexcept {
    const size_t w0 = nfull * (size_t)t / nt, w1 = nfull * (size_t)(t + 1) / nt;
    uint64_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    size_t w = w0;
    uint64_t wt = (uint64_t)(nw - w0);           // weight of word w
    for (; w + 4 <= w1; w += 4, wt -= 4) {
      uint64_t v[4];
      memcpy(v, p + 8 * w, 32);                  // (unaligned-safe; compiles to vector loads)
      const uint64_t m0 = v[0] ^ (v[0] >> 32), m1 = v[1] ^ (v[1] >> 32), m2 = v[2] ^ (v[2] >> 32), m3 = v[3] ^ (v[3] >> 32);
      a0 += m0; a1 += m1; a2 += m2; a3 += m3;
      b0 += m0 * wt; b1 += m1 * (wt - 1); b2 += m2 * (wt - 2); b3 += m3 * (wt - 3);
    }
    for (; w < w1; ++w, --wt) { uint64_t v; memcpy(&v, p + 8 * w, 8); const uint64_t m = v ^ (v >> 32); a0 += m; b0 += m * wt; }
    part[2 * t] = a0 + a1 + a2 + a3;
    part[2 * t + 1] = b0 + b1 + b2 + b3;
  };
  std::vector<std::thread> th;
  int started = 1;
  try {
    th.reserve((size_t)(nt - 1));
    for (int t = 1; t < nt; ++t) { th.emplace_back(run, t); ++started; }
  } catch (...) {
    // fewer helpers than planned: the calling thread does their slices below
  }
  run(0);
  for (int t = started; t < nt; ++t) run(t);
  for (auto& x : th) x.join();
  uint64_t A = 0, B = 0;
  for (int t = 0; t < nt; ++t) { A += part[2 * t]; B += part[2 * t + 1]; }
  if (nw > nfull) {                               // the last nbytes % 8 bytes, zero-extended: word nw - 1, weight 1
    uint64_t tail = 0;
    memcpy(&tail, p + 8 * nfull, nbytes - 8 * nfull);
    const uint64_t m = tail ^ (tail >> 32);
    A += m; B += m;
  }
  out[0] = A;
  out[1] = B;
  return MIMO_OK;
  });
}

double mimo_host_digamma(double x) { return digamma(x); }

int mimo_host_debug_fault(int kind) {
  return guarded_host([&]() -> int {
    if (kind == 1) throw std::bad_alloc();
    if (kind == 2) throw std::runtime_error("mimo_host_debug_fault");
    if (kind == 3) throw 42;
    if (kind == 4) { g_fail_thread_start.store(1); return MIMO_OK; }
    return kind == 0 ? MIMO_OK : MIMO_E_INVALID;
  });
}

}  // extern "C"
