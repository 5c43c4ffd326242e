// Microbenchmark: float64 issue rates on gfx950 that DESIGN.md prices the roofline against.
//   mfma  : v_mfma_f64_16x16x4_f64 on NACC independent accumulators (inline asm, tight loop)
//   valu  : v_fma_f64 on 8 independent chains
//   mixed : MFMA-only waves and VALU-only waves co-resident on every SIMD (do the pipes overlap?)
//   inter : MFMA and VALU instructions interleaved in ONE wave
// The in-kernel shader clock is s_memtime / s_memrealtime (100 MHz) over the loop.
// Build: hipcc --offload-arch=gfx950 -O3 tools/f64_rates.hip -o tools/f64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));

#define MFMA(acc, a, b) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define FMA(x, b, c) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c))

struct Stamp { unsigned long long t0, t1, r0, r1; };

__device__ inline void stamp_begin(Stamp& s) { s.t0 = __builtin_amdgcn_s_memtime(); s.r0 = __builtin_amdgcn_s_memrealtime(); }
__device__ inline void stamp_end(Stamp& s, Stamp* out) {
  s.t1 = __builtin_amdgcn_s_memtime(); s.r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

template <int NACC>
__device__ inline double mfma_loop(int iters, double a, double b) {
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) MFMA(acc[i], a, b);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  return s;
}

__device__ inline double valu_loop(int iters, double a0, double b0) {
  double x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = a0 * i + threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) FMA(x[i], b0, a0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  return s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, Stamp* st, int iters, double a0, double b0) {
  Stamp s; stamp_begin(s);
  double r = mfma_loop<NACC>(iters, a0 + threadIdx.x * 1e-9, b0);
  stamp_end(s, st);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

__global__ __launch_bounds__(256) void k_valu(double* out, Stamp* st, int iters, double a0, double b0) {
  Stamp s; stamp_begin(s);
  double r = valu_loop(iters, a0, b0);
  stamp_end(s, st);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// 512 threads: waves 0-3 MFMA (4 acc), waves 4-7 VALU; vi = VALU iterations (8 FMA each)
__global__ __launch_bounds__(512) void k_mixed(double* out, Stamp* st, int mi, int vi, double a0, double b0) {
  Stamp s; stamp_begin(s);
  double r;
  if ((threadIdx.x >> 6) < 4) r = mfma_loop<4>(mi, a0 + threadIdx.x * 1e-9, b0);
  else r = valu_loop(vi, a0, b0);
  stamp_end(s, st);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// one wave: per iteration 4 MFMA + NV v_fma_f64 interleaved
template <int NV>
__global__ __launch_bounds__(256) void k_inter(double* out, Stamp* st, int iters, double a0, double b0) {
  Stamp s; stamp_begin(s);
  d4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = a0 * i + threadIdx.x * 1e-9;
  double a = a0 + threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      MFMA(acc[i], a, b0);
#pragma unroll
      for (int v = 0; v < NV / 4; ++v) FMA(x[(i * (NV / 4) + v) & 7], b0, a0);
    }
  }
  double r = 0;
  for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) r += x[i];
  stamp_end(s, st);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static Stamp* d_st; static std::vector<Stamp> h_st;
static double clock_ghz(int grid) {
  hipMemcpy(h_st.data(), d_st, sizeof(Stamp) * grid, hipMemcpyDeviceToHost);
  std::vector<double> g;
  for (int i = 0; i < grid; ++i) {
    double dt = double(h_st[i].t1 - h_st[i].t0), dr = double(h_st[i].r1 - h_st[i].r0);
    if (dr > 0) g.push_back(dt / dr * 0.1);
  }
  std::sort(g.begin(), g.end());
  return g.empty() ? 0 : g[g.size() / 2];
}

template <typename F>
static float time_ms(F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) launch();
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("device %s, %d CUs, max clock %.0f MHz\n", p.gcnArchName, cus, p.clockRate / 1e3);
  double* out; hipMalloc(&out, sizeof(double) * cus * 8 * 512);
  hipMalloc(&d_st, sizeof(Stamp) * cus * 8); h_st.resize(cus * 8);
  const int iters = 40000;
  for (int w = 0; w < 100; ++w) hipLaunchKernelGGL(k_mfma<4>, dim3(cus * 2), dim3(256), 0, 0, out, d_st, iters, 1.0, 1e-3);
  hipDeviceSynchronize();
  auto rep_mfma = [&](const char* name, int nacc, int wpc, float ms) {
    const int grid = cus * wpc;
    const double ghz = clock_ghz(grid);
    const double n = double(iters) * nacc * wpc;  // MFMAs per SIMD
    printf("%-28s %d acc, %d wave/SIMD: %7.2f TFLOP/s  clock %.2f GHz  %.1f shader-cyc/MFMA/SIMD\n", name, nacc, wpc,
           2048.0 * iters * nacc * grid * 4 / ms / 1e9, ghz, ms * 1e-3 * ghz * 1e9 / n);
  };
  for (int wpc = 1; wpc <= 2; ++wpc) {
    const int grid = cus * wpc;
    float ms;
    ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<1>, dim3(grid), dim3(256), 0, 0, out, d_st, iters, 1.0, 1e-3); }); rep_mfma("mfma_f64_16x16x4", 1, wpc, ms);
    ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<2>, dim3(grid), dim3(256), 0, 0, out, d_st, iters, 1.0, 1e-3); }); rep_mfma("mfma_f64_16x16x4", 2, wpc, ms);
    ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(grid), dim3(256), 0, 0, out, d_st, iters, 1.0, 1e-3); }); rep_mfma("mfma_f64_16x16x4", 4, wpc, ms);
    ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<8>, dim3(grid), dim3(256), 0, 0, out, d_st, iters, 1.0, 1e-3); }); rep_mfma("mfma_f64_16x16x4", 8, wpc, ms);
    ms = time_ms([&] { hipLaunchKernelGGL(k_valu, dim3(grid), dim3(256), 0, 0, out, d_st, iters * 4, 1.0, 0.999); });
    {
      const double ghz = clock_ghz(grid);
      printf("%-28s 8 chains, %d wave/SIMD: %7.2f TFLOP/s  clock %.2f GHz  %.2f shader-cyc/FMA/SIMD\n", "v_fma_f64", wpc,
             128.0 * iters * 4 * 8 * grid * 4 / ms / 1e9, ghz, ms * 1e-3 * ghz * 1e9 / (double(iters) * 4 * 8 * wpc));
    }
  }
  // mixed: sweep the VALU share
  for (int vmul : {0, 2, 4, 8, 12, 16, 24}) {   // v_fma_f64 per MFMA (VALU waves vs MFMA waves)
    const int grid = cus, vi = vmul * iters / 2;   // 8 FMAs per VALU iteration, 4 MFMAs per MFMA iteration
    float ms = time_ms([&] { hipLaunchKernelGGL(k_mixed, dim3(grid), dim3(512), 0, 0, out, d_st, iters, vi, 1.0, 0.999); });
    const double ghz = clock_ghz(grid);
    const double mf = 2048.0 * iters * 4 * grid * 4, vf = 128.0 * double(vi) * 8 * grid * 4;
    printf("mixed waves: %2d FMA per MFMA: %.3f ms clock %.2f GHz  mfma %.2f + valu %.2f = %.2f TFLOP/s\n", vmul, ms, ghz,
           mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9);
  }
  {
    const int grid = cus * 2;
    float ms;
    ms = time_ms([&] { hipLaunchKernelGGL(k_inter<4>, dim3(grid), dim3(256), 0, 0, out, d_st, iters, 1.0, 0.999); });
    printf("interleaved 1 FMA/MFMA, 2 wave/SIMD: %.3f ms clock %.2f  mfma %.2f + valu %.2f TFLOP/s\n", ms, clock_ghz(grid),
           2048.0 * iters * 4 * grid * 4 / ms / 1e9, 128.0 * iters * 4 * grid * 4 / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_inter<16>, dim3(grid), dim3(256), 0, 0, out, d_st, iters, 1.0, 0.999); });
    printf("interleaved 4 FMA/MFMA, 2 wave/SIMD: %.3f ms clock %.2f  mfma %.2f + valu %.2f TFLOP/s\n", ms, clock_ghz(grid),
           2048.0 * iters * 4 * grid * 4 / ms / 1e9, 128.0 * iters * 16 * grid * 4 / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_inter<32>, dim3(grid), dim3(256), 0, 0, out, d_st, iters, 1.0, 0.999); });
    printf("interleaved 8 FMA/MFMA, 2 wave/SIMD: %.3f ms clock %.2f  mfma %.2f + valu %.2f TFLOP/s\n", ms, clock_ghz(grid),
           2048.0 * iters * 4 * grid * 4 / ms / 1e9, 128.0 * iters * 32 * grid * 4 / ms / 1e9);
  }
  hipFree(out);
  return 0;
}
