// Microbenchmark: float64 issue rates on gfx950 that the roofline in DESIGN.md is priced against.
//   mfma : back-to-back v_mfma_f64_16x16x4_f64 on NACC independent accumulators
//   valu : v_fma_f64 chains
//   both : one MFMA-only wave and one VALU-only wave per SIMD (do the pipes overlap?)
// Build: hipcc --offload-arch=gfx950 -O3 tools/f64_rates.hip -o tools/f64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_valu(double* out, int iters, double a0, double b0) {
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = a0 * i + threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = fma(x[i], b0, a0);
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 512 threads: waves 0-3 MFMA, waves 4-7 VALU (one of each per SIMD)
__global__ __launch_bounds__(512) void k_both(double* out, int iters, double a0, double b0) {
  const int wave = threadIdx.x >> 6;
  double s = 0;
  if (wave < 4) {
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = a0 * i + threadIdx.x * 1e-9;
    for (int it = 0; it < iters * 2; ++it) {   // 16 v_fma_f64 (=64 cyc if 4 cyc each) per 4 MFMAs
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = fma(x[i], b0, a0);
    }
    for (int i = 0; i < 8; ++i) s += x[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static float time_ms(F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("device %s, %d CUs, clock %.0f MHz\n", p.name, cus, p.clockRate / 1e3);
  double* out; hipMalloc(&out, sizeof(double) * cus * 8 * 512);
  const int iters = 20000;
  for (int wpc = 1; wpc <= 2; ++wpc) {   // workgroups per CU (waves per SIMD)
    const int grid = cus * wpc;
    float ms;
    ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-3); });
    printf("mfma f64 16x16x4, 1 acc, %d wave/SIMD: %.2f TFLOP/s  (%.1f cyc/MFMA/SIMD @2.4GHz)\n", wpc,
           2048.0 * iters * grid * 4 / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * wpc));
    ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-3); });
    printf("mfma f64 16x16x4, 4 acc, %d wave/SIMD: %.2f TFLOP/s  (%.1f cyc/MFMA/SIMD @2.4GHz)\n", wpc,
           2048.0 * iters * 4 * grid * 4 / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 4 * wpc));
    ms = time_ms([&] { hipLaunchKernelGGL(k_valu, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 0.999); });
    printf("valu v_fma_f64, 8 chains, %d wave/SIMD: %.2f TFLOP/s  (%.1f cyc/FMA/SIMD @2.4GHz)\n", wpc,
           128.0 * iters * 8 * grid * 4 / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 8 * wpc));
  }
  {
    const int grid = cus;
    float ms = time_ms([&] { hipLaunchKernelGGL(k_both, dim3(grid), dim3(512), 0, 0, out, iters, 1.0, 0.999); });
    const double mf = 2048.0 * iters * 4 * grid * 4, vf = 128.0 * iters * 2 * 8 * grid * 4;
    printf("both (1 MFMA wave + 1 VALU wave per SIMD): %.3f ms; mfma %.2f TF + valu %.2f TF = %.2f TFLOP/s\n", ms,
           mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9);
  }
  hipFree(out);
  return 0;
}
