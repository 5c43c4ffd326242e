// Microbenchmark: issue cost (shader cycles per wave-instruction per SIMD) of the VALU instructions the normalise
// phases are made of, at 1 / 2 / 4 / 8 waves per SIMD, 8 independent chains per wave: v_fma_f64, v_add_f64, v_mul_f64,
// v_max_f64, v_fma_f32, v_and_b32 / v_lshl_add_u32 (integer), and a 1 : 1 mix of f64 and integer instructions.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o tools/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>

#define OP8(fmt)                                                                                                  \
  asm volatile(fmt "\n\t" fmt##1 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b))

template <int KIND>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a, double b) {
  double x0 = a * 1 + threadIdx.x * 1e-9, x1 = a * 2, x2 = a * 3, x3 = a * 4, x4 = a * 5, x5 = a * 6, x6 = a * 7, x7 = a * 8;
  int i0 = threadIdx.x, i1 = 1, i2 = 2, i3 = 3, i4 = 4, i5 = 5, i6 = 6, i7 = 7, ia = 0x7fffffff, ib = 3;
  float f0 = threadIdx.x, f1 = 1, f2 = 2, f3 = 3, f4 = 4, f5 = 5, f6 = 6, f7 = 7, fa = 0.999f, fb = 1e-3f;
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 0)
      asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                   "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(a));
    if constexpr (KIND == 1)
      asm volatile("v_add_f64 %0, %0, %8\n\tv_add_f64 %1, %1, %8\n\tv_add_f64 %2, %2, %8\n\tv_add_f64 %3, %3, %8\n\t"
                   "v_add_f64 %4, %4, %8\n\tv_add_f64 %5, %5, %8\n\tv_add_f64 %6, %6, %8\n\tv_add_f64 %7, %7, %8"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
    if constexpr (KIND == 2)
      asm volatile("v_mul_f64 %0, %0, %8\n\tv_mul_f64 %1, %1, %8\n\tv_mul_f64 %2, %2, %8\n\tv_mul_f64 %3, %3, %8\n\t"
                   "v_mul_f64 %4, %4, %8\n\tv_mul_f64 %5, %5, %8\n\tv_mul_f64 %6, %6, %8\n\tv_mul_f64 %7, %7, %8"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));
    if constexpr (KIND == 3)
      asm volatile("v_max_f64 %0, %0, %8\n\tv_max_f64 %1, %1, %8\n\tv_max_f64 %2, %2, %8\n\tv_max_f64 %3, %3, %8\n\t"
                   "v_max_f64 %4, %4, %8\n\tv_max_f64 %5, %5, %8\n\tv_max_f64 %6, %6, %8\n\tv_max_f64 %7, %7, %8"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
    if constexpr (KIND == 4)
      asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                   "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fa), "v"(fb));
    if constexpr (KIND == 5)
      asm volatile("v_lshl_add_u32 %0, %0, 1, %8\n\tv_and_b32 %1, %1, %9\n\tv_lshl_add_u32 %2, %2, 1, %8\n\tv_and_b32 %3, %3, %9\n\t"
                   "v_lshl_add_u32 %4, %4, 1, %8\n\tv_and_b32 %5, %5, %9\n\tv_lshl_add_u32 %6, %6, 1, %8\n\tv_and_b32 %7, %7, %9"
                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(ib), "v"(ia));
    if constexpr (KIND == 6)      // 4 f64 + 4 integer, alternating
      asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_lshl_add_u32 %4, %4, 1, %10\n\tv_fma_f64 %1, %1, %8, %9\n\tv_and_b32 %5, %5, %11\n\t"
                   "v_fma_f64 %2, %2, %8, %9\n\tv_lshl_add_u32 %6, %6, 1, %10\n\tv_fma_f64 %3, %3, %8, %9\n\tv_and_b32 %7, %7, %11"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(b), "v"(a), "v"(ib), "v"(ia));
    if constexpr (KIND == 7)      // dependent chain of v_fma_f64 (latency)
      asm volatile("v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\t"
                   "v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2"
                   : "+v"(x0) : "v"(b), "v"(a));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}

template <int KIND>
static void run(const char* name, int cus, double* out) {
  const int iters = 20000;
  printf("%-34s", name);
  for (int wps : {1, 2, 4, 8}) {
    const int grid = cus * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, iters, 0.999, 1e-3);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, iters, 0.999, 1e-3); hipEventRecord(e1);
      hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    // cycles per wave-instruction per SIMD at 2.4 GHz nominal
    printf("  %dw: %5.2f", wps, best * 1e-3 * 2.4e9 / (double(iters) * 8 * wps));
  }
  printf("   (cycles at 2.4 GHz per instruction per SIMD)\n");
}

// co-residency: waves 0-3 of a 512-thread workgroup issue v_mfma_f64_16x16x4_f64 back to back, waves 4-7 a VALU stream
// of one kind (0: v_fma_f64, 4: v_fma_f32, 5: integer).  Do the VALU instructions overlap the matrix instructions?
typedef double d4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(512) void kmix(double* out, int mi, int vi, double a, double b) {
  double r = 0;
  if ((threadIdx.x >> 6) < 4) {
    d4 acc[4] = {d4{0, 0, 0, 0}, d4{0, 0, 0, 0}, d4{0, 0, 0, 0}, d4{0, 0, 0, 0}};
    double av = a + threadIdx.x * 1e-9;
    for (int it = 0; it < mi; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(b));
    }
    for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    double x0 = a, x1 = a * 2, x2 = a * 3, x3 = a * 4;
    float f0 = 1, f1 = 2, f2 = 3, f3 = 4, fa = 0.999f, fb = 1e-3f;
    int i0 = threadIdx.x, i1 = 1, i2 = 2, i3 = 3, ia = 0x7fffffff, ib = 3;
    for (int it = 0; it < vi; ++it) {
      if constexpr (KIND == 0)
        asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b), "v"(a));
      if constexpr (KIND == 4)
        asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                     : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fa), "v"(fb));
      if constexpr (KIND == 5)
        asm volatile("v_lshl_add_u32 %0, %0, 1, %4\n\tv_and_b32 %1, %1, %5\n\tv_lshl_add_u32 %2, %2, 1, %4\n\tv_and_b32 %3, %3, %5"
                     : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(ib), "v"(ia));
    }
    r = x0 + x1 + x2 + x3 + f0 + f1 + f2 + f3 + i0 + i1 + i2 + i3;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
static void runmix(const char* name, int cus, double* out) {
  const int mi = 20000;      // 4 MFMAs per iteration: 80000 MFMAs = 5.12e6 cycles alone
  printf("%-22s", name);
  for (int per : {0, 4, 8, 16}) {      // VALU instructions per MFMA
    const int vi = per * mi;             // 4 VALU per iteration
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kmix<KIND>, dim3(cus), dim3(512), 0, 0, out, mi, vi, 0.999, 1e-3);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0); hipLaunchKernelGGL(kmix<KIND>, dim3(cus), dim3(512), 0, 0, out, mi, vi, 0.999, 1e-3); hipEventRecord(e1);
      hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    printf("  %2d/MFMA: %6.3f ms", per, best);
  }
  printf("\n");
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double* out; hipMalloc(&out, sizeof(double) * cus * 8 * 256);
  run<0>("v_fma_f64 (8 chains)", cus, out);
  run<1>("v_add_f64", cus, out);
  run<2>("v_mul_f64", cus, out);
  run<3>("v_max_f64", cus, out);
  run<4>("v_fma_f32", cus, out);
  run<5>("v_lshl_add_u32 / v_and_b32", cus, out);
  run<6>("f64 fma : int 1 : 1", cus, out);
  run<7>("v_fma_f64 dependent chain", cus, out);
  printf("MFMA f64 waves + VALU waves on the same SIMDs (time of 80000 MFMAs per SIMD alone: first column)\n");
  runmix<0>("with v_fma_f64", cus, out);
  runmix<4>("with v_fma_f32", cus, out);
  runmix<5>("with integer VALU", cus, out);
  return 0;
}
