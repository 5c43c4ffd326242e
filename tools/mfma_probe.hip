// Microbenchmark + layout probe for v_mfma_f64_4x4x4_4b_f64 on gfx950:
//   layout : one-hot A (lane la) x one-hot B (lane lb) -> which D lanes light up; prints the (block, i, k) / (block, k, j) /
//            (block, i, j) decomposition of the lane index for A, B and D
//   rate   : issue interval with NACC independent accumulators at 1 / 2 / 4 waves per SIMD, next to 16x16x4
//   chain  : dependent chain (one accumulator) latency
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(double* out) {      // out[la][lb][64]
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      out[((size_t)la * 64 + lb) * 64 + lane] = d;
    }
}

#define MFMA4(acc, a, b) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA16(acc, a, b) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define FMA(x, b, c) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c))

struct Stamp { unsigned long long t0, t1, r0, r1; };

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma4(double* out, Stamp* st, int iters, double a0, double b0) {
  Stamp s; s.t0 = __builtin_amdgcn_s_memtime(); s.r0 = __builtin_amdgcn_s_memrealtime();
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  const double a = a0 + threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) MFMA4(acc[i], a, b0);
  }
  double r = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) r += acc[i];
  s.t1 = __builtin_amdgcn_s_memtime(); s.r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) st[blockIdx.x] = s;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// NV v_fma_f64 (independent chains) per 4x4x4 MFMA in ONE wave
template <int NV>
__global__ __launch_bounds__(256) void k_inter4(double* out, Stamp* st, int iters, double a0, double b0) {
  Stamp s; s.t0 = __builtin_amdgcn_s_memtime(); s.r0 = __builtin_amdgcn_s_memrealtime();
  double acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = 0.0;
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = a0 * i + threadIdx.x * 1e-9;
  const double a = a0 + threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      MFMA4(acc[i], a, b0);
#pragma unroll
      for (int v = 0; v < NV; ++v) FMA(x[(i * NV + v) & 7], b0, a0);
    }
  }
  double r = 0;
  for (int i = 0; i < 8; ++i) r += acc[i] + x[i];
  s.t1 = __builtin_amdgcn_s_memtime(); s.r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) st[blockIdx.x] = s;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

__global__ __launch_bounds__(256) void k_valu(double* out, Stamp* st, int iters, double a0, double b0) {
  Stamp s; s.t0 = __builtin_amdgcn_s_memtime(); s.r0 = __builtin_amdgcn_s_memrealtime();
  double x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = a0 * i + threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) FMA(x[i], b0, a0);
  }
  double r = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += x[i];
  s.t1 = __builtin_amdgcn_s_memtime(); s.r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) st[blockIdx.x] = s;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// bursts, as narrow_kernel issues them: every wave alternates NB 4x4x4 MFMAs (independent accumulators) with NVAL independent
// v_fma_f64; PRIO = 1: the MFMA burst at priority 3; waves of a SIMD start at different phases (wave index)
template <int NB, int NVAL, int PRIO>
__global__ __launch_bounds__(256) void k_burst(double* out, Stamp* st, int iters, double a0, double b0) {
  Stamp s; s.t0 = __builtin_amdgcn_s_memtime(); s.r0 = __builtin_amdgcn_s_memrealtime();
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = 0.0;
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = a0 * i + threadIdx.x * 1e-9;
  const double a = a0 + threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
    if (PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int i = 0; i < NB; ++i) MFMA4(acc[i & 15], a, b0);
    if (PRIO) __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int v = 0; v < NVAL; ++v) FMA(x[v & 7], b0, a0);
  }
  double r = 0;
  for (int i = 0; i < 16; ++i) r += acc[i];
  for (int i = 0; i < 8; ++i) r += x[i];
  s.t1 = __builtin_amdgcn_s_memtime(); s.r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) st[blockIdx.x] = s;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// 4x4x4 MFMAs whose B operand comes from LDS, as narrow_kernel's first product: DEPTH ds_read_b64 in flight, every read feeds
// NM MFMAs; PATTERN 0: the 16 distinct 8-byte words of a slice, replicated over the four blocks (lane bits 2-3) — the narrow
// kernel's read; 1: 64 distinct words (no broadcast)
template <int DEPTH, int NM, int PATTERN>
__global__ __launch_bounds__(256) void k_ldsfeed(double* out, Stamp* st, int iters, double a0) {
  __shared__ double tab[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 256) tab[i] = 1e-3 * i;
  __syncthreads();
  Stamp s; s.t0 = __builtin_amdgcn_s_memtime(); s.r0 = __builtin_amdgcn_s_memrealtime();
  const int lane = threadIdx.x & 63;
  const double* p = tab + (PATTERN == 0 ? 4 * (lane >> 4) + (lane & 3) : lane);
  constexpr int STR = PATTERN == 0 ? 16 : 64;
  double acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = 0.0;
  const double a = a0 + threadIdx.x * 1e-9;
  double ring[DEPTH];
#pragma unroll
  for (int e = 0; e < DEPTH; ++e) ring[e] = p[e * STR];
  for (int it = 0; it < iters; ++it) {
    asm volatile("" ::: "memory");       // the table reads stay in the loop
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      const double tv = ring[e % DEPTH];
      ring[e % DEPTH] = p[((e + DEPTH) & 31) * STR];
#pragma unroll
      for (int j = 0; j < NM; ++j) acc[(e * NM + j) & 7] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, tv, acc[(e * NM + j) & 7], 0, 0, 0);
    }
  }
  double r = 0;
  for (int i = 0; i < 8; ++i) r += acc[i];
  s.t1 = __builtin_amdgcn_s_memtime(); s.r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) st[blockIdx.x] = s;
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
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("device %s, %d CUs\n", p.gcnArchName, cus);
  // ---- layout ------------------------------------------------------------------------------
  {
    double* d; hipMalloc(&d, sizeof(double) * 64 * 64 * 64);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, d);
    std::vector<double> h(64 * 64 * 64);
    hipMemcpy(h.data(), d, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
    // for every A lane: the set of B lanes it meets, and the D lane of every meeting
    printf("v_mfma_f64_4x4x4_4b_f64 layout: for A lane la, the B lanes lb with a non-zero product and the D lane that receives it\n");
    for (int la = 0; la < 64; ++la) {
      printf("  A lane %2d:", la);
      for (int lb = 0; lb < 64; ++lb)
        for (int l = 0; l < 64; ++l)
          if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0) printf(" (B %2d -> D %2d)", lb, l);
      printf("\n");
    }
    hipFree(d);
  }
  // ---- rates -------------------------------------------------------------------------------
  double* out; hipMalloc(&out, sizeof(double) * cus * 16 * 256);
  hipMalloc(&d_st, sizeof(Stamp) * cus * 16); h_st.resize(cus * 16);
  const int iters = 40000;
  for (int w = 0; w < 50; ++w) hipLaunchKernelGGL(k_mfma4<4>, dim3(cus * 2), dim3(256), 0, 0, out, d_st, iters, 1.0, 1e-3);
  hipDeviceSynchronize();
  auto rep = [&](int nacc, int wpc, float ms) {
    const int grid = cus * wpc;
    const double ghz = clock_ghz(grid);
    const double n = double(iters) * nacc * wpc;
    printf("mfma_f64_4x4x4_4b  %d acc, %d wave/SIMD: %7.2f TFLOP/s  clock %.2f GHz  %.1f shader-cyc/MFMA/SIMD\n", nacc, wpc,
           512.0 * iters * nacc * grid * 4 / ms / 1e9, ghz, ms * 1e-3 * ghz * 1e9 / n);
  };
  for (int wpc : {1, 2, 4}) {
    const int grid = cus * wpc;
    float ms;
    ms = time_ms([&] { hipLaunchKernelGGL(k_mfma4<1>, dim3(grid), dim3(256), 0, 0, out, d_st, iters, 1.0, 1e-3); }); rep(1, wpc, ms);
    ms = time_ms([&] { hipLaunchKernelGGL(k_mfma4<2>, dim3(grid), dim3(256), 0, 0, out, d_st, iters, 1.0, 1e-3); }); rep(2, wpc, ms);
    ms = time_ms([&] { hipLaunchKernelGGL(k_mfma4<4>, dim3(grid), dim3(256), 0, 0, out, d_st, iters, 1.0, 1e-3); }); rep(4, wpc, ms);
    ms = time_ms([&] { hipLaunchKernelGGL(k_mfma4<8>, dim3(grid), dim3(256), 0, 0, out, d_st, iters, 1.0, 1e-3); }); rep(8, wpc, ms);
  }
  for (int wpc : {1, 2, 4, 8}) {
    const int grid = cus * wpc;
    float ms = time_ms([&] { hipLaunchKernelGGL(k_valu, dim3(grid), dim3(256), 0, 0, out, d_st, iters * 2, 1.0, 0.999); });
    printf("v_fma_f64 8 chains, %d wave/SIMD: %7.2f TFLOP/s  clock %.2f GHz  %.2f shader-cyc/FMA/SIMD\n", wpc,
           128.0 * iters * 2 * 8 * grid * 4 / ms / 1e9, clock_ghz(grid), ms * 1e-3 * clock_ghz(grid) * 1e9 / (double(iters) * 2 * 8 * wpc));
  }
  for (int wpc : {2, 4}) {
    const int grid = cus * wpc;
    float ms;
    ms = time_ms([&] { hipLaunchKernelGGL(k_inter4<1>, dim3(grid), dim3(256), 0, 0, out, d_st, iters, 1.0, 0.999); });
    printf("interleaved 1 FMA per 4x4x4 MFMA, %d wave/SIMD: %.1f shader-cyc per (MFMA + 1 FMA) per SIMD\n", wpc, ms * 1e-3 * clock_ghz(grid) * 1e9 / (double(iters) * 8 * wpc));
    ms = time_ms([&] { hipLaunchKernelGGL(k_inter4<2>, dim3(grid), dim3(256), 0, 0, out, d_st, iters, 1.0, 0.999); });
    printf("interleaved 2 FMA per 4x4x4 MFMA, %d wave/SIMD: %.1f shader-cyc per (MFMA + 2 FMA) per SIMD\n", wpc, ms * 1e-3 * clock_ghz(grid) * 1e9 / (double(iters) * 8 * wpc));
    ms = time_ms([&] { hipLaunchKernelGGL(k_inter4<4>, dim3(grid), dim3(256), 0, 0, out, d_st, iters, 1.0, 0.999); });
    printf("interleaved 4 FMA per 4x4x4 MFMA, %d wave/SIMD: %.1f shader-cyc per (MFMA + 4 FMA) per SIMD\n", wpc, ms * 1e-3 * clock_ghz(grid) * 1e9 / (double(iters) * 8 * wpc));
  }
  {
    const int it2 = 2000;
    auto rep2 = [&](const char* what, int nb, int nval, int wpc, float ms) {
      const int grid = cus * wpc;
      const double cyc = ms * 1e-3 * clock_ghz(grid) * 1e9 / (double(it2) * wpc);
      printf("bursts of %d MFMA(4x4x4) + %d v_fma_f64 per wave, %s, %d wave/SIMD: %.0f cycles per (burst pair) per SIMD; sum of the issue times %.0f\n",
             nb, nval, what, wpc, cyc, nb * 16.2 + nval * 4.4);
    };
    for (int wpc : {1, 2, 3, 4}) {
      const int grid = cus * wpc;
      float ms;
      ms = time_ms([&] { hipLaunchKernelGGL((k_burst<32, 128, 0>), dim3(grid), dim3(256), 0, 0, out, d_st, it2, 1.0, 0.999); }); rep2("no priorities", 32, 128, wpc, ms);
      ms = time_ms([&] { hipLaunchKernelGGL((k_burst<32, 128, 1>), dim3(grid), dim3(256), 0, 0, out, d_st, it2, 1.0, 0.999); }); rep2("MFMA at priority 3", 32, 128, wpc, ms);
      ms = time_ms([&] { hipLaunchKernelGGL((k_burst<64, 256, 0>), dim3(grid), dim3(256), 0, 0, out, d_st, it2, 1.0, 0.999); }); rep2("no priorities", 64, 256, wpc, ms);
      ms = time_ms([&] { hipLaunchKernelGGL((k_burst<0, 256, 0>), dim3(grid), dim3(256), 0, 0, out, d_st, it2, 1.0, 0.999); }); rep2("VALU only", 0, 256, wpc, ms);
      ms = time_ms([&] { hipLaunchKernelGGL((k_burst<64, 0, 0>), dim3(grid), dim3(256), 0, 0, out, d_st, it2, 1.0, 0.999); }); rep2("MFMA only", 64, 0, wpc, ms);
    }
  }
  {
    const int it3 = 2000;
    auto rep3 = [&](int depth, int nm, int pat, int wpc, float ms) {
      const int grid = cus * wpc;
      printf("LDS-fed 4x4x4 MFMA: %d reads in flight, %d MFMA per read, %s, %d wave/SIMD: %.1f cycles per MFMA per SIMD\n", depth, nm,
             pat ? "64 distinct words" : "16 words x 4 broadcast", wpc, ms * 1e-3 * clock_ghz(grid) * 1e9 / (double(it3) * 32 * nm * wpc));
    };
    for (int wpc : {1, 2, 3, 4}) {
      const int grid = cus * wpc;
      float ms;
#define LF(D_, NM_, P_) ms = time_ms([&] { hipLaunchKernelGGL((k_ldsfeed<D_, NM_, P_>), dim3(grid), dim3(256), 0, 0, out, d_st, it3, 1.0); }); rep3(D_, NM_, P_, wpc, ms);
      LF(2, 1, 0) LF(4, 1, 0) LF(8, 1, 0) LF(16, 1, 0) LF(8, 1, 1) LF(4, 2, 0) LF(8, 2, 0) LF(4, 4, 0)
#undef LF
    }
  }
  hipFree(out);
  return 0;
}
