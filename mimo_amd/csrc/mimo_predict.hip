// Posterior-predictive mixture moments for narrow inputs (mimo_predict; reference: mimo/mixtures/ilr.py:374-430 meanfield_prediction,
// bayesian.py:949-962 posterior_predictive_gaussian, utils/stats.py:53-66): the register form of predict_kernel (mimo_kernels.hip).
//
// The generic kernel keeps the row x~ in a thread-private LDS column because dx is a run-time number: every multiply-add of the
// quadratic forms reads one factor from LDS — one LDS instruction per fma, 6 - 10 % of the float64 vector rate (profiles/
// r03_predict_kernel_device_resident.txt).  Here dx and dy are template parameters (dx <= 8 with the affine column: the shapes of the
// reference's ILR examples — dx = dy = 1 in evaluate_sine / chirp / sinc, the 8 -> 4 shape of BASELINE config 4): x~ lives in
// registers, every loop is unrolled, the per-component blocks arrive through scalar (uniform-address) loads and enter the fmas as
// scalar operands, and the two exponentials per component are the table-driven exp_nonpos (15 instructions instead of libm's ~40).
// The blocks are read through the CONSTANT address space (they do not change while the kernel runs): hipcc then issues s_load
// through the scalar cache — as plain global pointers the uniform-address loads were vector memory instructions with a wait
// right behind them, and the component loop ran at L2 latency (474 cycles per component and wave at dx = dy = 1 for ~60 instructions).
// One thread per row; the softmax takes two passes over the components (maximum of the gate values first — dx^2 + dx fmas per
// component —, then ONE exponential per component instead of the two of an online softmax: at dx = dy = 1 the exponentials are
// two thirds of the instructions); arg-max mode and the predictive log-density as in the generic kernel.
#include "mimo_device.h"

namespace mimo {

typedef const double __attribute__((address_space(4)))* cptr_t;       // constant address space: uniform loads become s_load

template <int DY, int DX>
__global__ __launch_bounds__(256) void predict_reg_kernel(const PredictArgs a) {
  constexpr int DC = DX + 1;                   // affine: x~ = [x, 1]
  __shared__ double etab[kExpTab];
  const int tid = threadIdx.x;
  for (int e = tid; e < kExpTab; e += 256) etab[e] = exp_tab_entry_c(e);
  wg_sync();
  const cptr_t gate_p = (cptr_t)a.gate, M_p = (cptr_t)a.M, Q_p = (cptr_t)a.Q, Cc_p = (cptr_t)a.Cc;
  const int64_t n = (int64_t)blockIdx.x * 256 + tid;
  const bool valid = n < a.N;
  const int K = a.K;
  double x[DC];
#pragma unroll
  for (int i = 0; i < DX; ++i) x[i] = valid ? a.Z[n * DX + i] : 0.0;
  x[DX] = 1.0;

  auto gate = [&](int k) {
    const cptr_t t = gate_p + (size_t)k * (1 + DX + DX * DX);
    double l = t[0];
#pragma unroll
    for (int i = 0; i < DX; ++i) {
      double q = 0.0;
#pragma unroll
      for (int j = 0; j < DX; ++j) q = fma(t[1 + DX + i * DX + j], x[j], q);
      l = fma(x[i], t[1 + i] - 0.5 * q, l);
    }
    return l;
  };
  auto expert = [&](int k, double (&m)[DY], double& cs) {
    const cptr_t Mk = M_p + (size_t)k * DY * DC;
    const cptr_t Qk = Q_p + (size_t)k * DC * DC;
#pragma unroll
    for (int d = 0; d < DY; ++d) m[d] = 0.0;
    double q = 0.0;
#pragma unroll
    for (int i = 0; i < DC; ++i) {
      double r = 0.0;
#pragma unroll
      for (int j = 0; j < DC; ++j) r = fma(Qk[i * DC + j], x[j], r);
      q = fma(x[i], r, q);
#pragma unroll
      for (int d = 0; d < DY; ++d) m[d] = fma(Mk[d * DC + i], x[i], m[d]);
    }
    cs = 1.0 + q;
  };

  double mx = -1e308, ssum = 0.0;
  double amu[DY], aS[DY][DY];
#pragma unroll
  for (int d = 0; d < DY; ++d) {
    amu[d] = 0.0;
#pragma unroll
    for (int e = 0; e < DY; ++e) aS[d][e] = 0.0;
  }
  int best = 0;
  if (a.mode == 1) {            // argmax only (first maximum, as np.argmax)
    for (int k = 0; k < K; ++k) {
      const double l = gate(k);
      if (l > mx) { mx = l; best = k; }
    }
  } else {
    for (int k = 0; k < K; ++k) mx = fmax(mx, gate(k));
    for (int k = 0; k < K; ++k) {
      const double w = exp_nonpos_t2048c(gate(k) - mx, etab);
      double m[DY], cs;
      expert(k, m, cs);
      const cptr_t Ck = Cc_p + (size_t)k * DY * DY;
      ssum += w;
#pragma unroll
      for (int d = 0; d < DY; ++d) {
        amu[d] = fma(w, m[d], amu[d]);
#pragma unroll
        for (int e = 0; e < DY; ++e) aS[d][e] = fma(w, fma(cs, Ck[d * DY + e], m[d] * m[e]), aS[d][e]);
      }
    }
  }
  if (a.mode == 1) {
    double m[DY], cs;
    expert(best, m, cs);
    const cptr_t Ck = Cc_p + (size_t)best * DY * DY;
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
      for (int k = 0; k < K; ++k) ssum += exp_nonpos_t2048c(gate(k) - mx, etab);
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
    const double lse = mx + log(ssum);
    double yv[DY];
#pragma unroll
    for (int d = 0; d < DY; ++d) yv[d] = valid ? a.y[n * DY + d] : 0.0;
    double tm = -INFINITY, ts = 0.0;
    for (int k = 0; k < K; ++k) {
      const double w = exp(gate(k) - lse);
      double m[DY], cs;
      expert(k, m, cs);
      const cptr_t Pk = (cptr_t)a.P + (size_t)k * DY * DY;
      double q = 0.0;
#pragma unroll
      for (int d = 0; d < DY; ++d) {
        double r = 0.0;
#pragma unroll
        for (int e = 0; e < DY; ++e) r = fma(Pk[d * DY + e], yv[e] - m[e], r);
        q = fma(yv[d] - m[d], r, q);
      }
      const double lpl = -0.5 * q / cs - 0.5 * DY * 1.8378770664093453 + 0.5 * (((cptr_t)a.ld)[k] - DY * log(cs));
      const double t = lpl + log(w + 2.2250738585072014e-308);
      if (t > tm) { ts = ts * exp(tm - t) + 1.0; tm = t; }
      else ts += exp(t - tm);
    }
    if (valid) a.nlpd[n] = -(tm + log(ts));
  }
}

typedef void (*predict_fn)(const PredictArgs);
template <int DX>
static predict_fn pick_predict_dy(int dy) {
  switch (dy) {
    case 1: return predict_reg_kernel<1, DX>; case 2: return predict_reg_kernel<2, DX>; case 3: return predict_reg_kernel<3, DX>;
    case 4: return predict_reg_kernel<4, DX>; case 5: return predict_reg_kernel<5, DX>; case 6: return predict_reg_kernel<6, DX>;
    case 7: return predict_reg_kernel<7, DX>; case 8: return predict_reg_kernel<8, DX>;
  }
  return nullptr;
}
// true: the register kernel took the call (*err = launch status); false: shape outside its range (dx > 8, no affine column,
// MIMO_PREDICT_REG=0) — the generic kernel runs
bool launch_predict_reg(const PredictArgs& a, hipStream_t stream, hipError_t* err) {
  static const bool on = [] { const char* e = getenv("MIMO_PREDICT_REG"); return !e || atoi(e) != 0; }();   // tuning knob
  if (!on || a.dc != a.dx + 1 || a.dx < 1 || a.dx > 8 || a.dy < 1 || a.dy > kMaxPredictDy) return false;
  predict_fn fn = nullptr;
  switch (a.dx) {
    case 1: fn = pick_predict_dy<1>(a.dy); break; case 2: fn = pick_predict_dy<2>(a.dy); break;
    case 3: fn = pick_predict_dy<3>(a.dy); break; case 4: fn = pick_predict_dy<4>(a.dy); break;
    case 5: fn = pick_predict_dy<5>(a.dy); break; case 6: fn = pick_predict_dy<6>(a.dy); break;
    case 7: fn = pick_predict_dy<7>(a.dy); break; case 8: fn = pick_predict_dy<8>(a.dy); break;
  }
  if (!fn) return false;
  *err = hipSuccess;
  if (a.N <= 0) return true;
  hipLaunchKernelGGL(fn, dim3((unsigned)((a.N + 255) / 256)), dim3(256), 0, stream, a);
  *err = hipGetLastError();
  return true;
}

}  // namespace mimo
