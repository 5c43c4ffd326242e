// Device-side helpers shared by the gfx950 kernel files (mimo_kernels.hip: MFMA tile kernels; mimo_small.hip: the
// VALU kernel for tiny K x F; mimo_rowwave.hip: the large-K Gibbs label kernel and the label-statistics pass).
#pragma once
#include "mimo_kernels.h"

#include <math.h>

namespace mimo {

// Workgroup barrier with the wave's own LDS traffic drained first.  hipcc places `s_waitcnt lgkmcnt(0)` in front
// of an s_barrier only where its memory model asks for it, and for LDS-only ordering at workgroup scope it does
// not: LLVM assumes the LDS operations of all waves execute in ONE total order, so stores issued before the
// barrier would be seen by loads other waves issue after it.  On gfx950 with two workgroups resident per CU that
// does not hold: a wave signalled the loop-top barrier with its z-tile stores still queued (behind the co-resident
// workgroup's bank-conflicted traffic), the other waves built the feature tile from the PREVIOUS tile's rows, and
// 2-8 rows of a tile came out wrong — sporadically, only on tiles after a workgroup's first, only with two
// workgroups per CU (found with a lane-layout experiment that made the window wide: 35 of 48 stress runs bad,
// 0 of 120 with this wait; DESIGN.md section 4 has the story).  88 of the 179
// kernel instantiations had such a barrier (tools/check_barrier_waits.py, a CFG dataflow over the emitted ISA,
// now part of the CPU tests).  The explicit wait costs nothing measurable: most barriers had it already.
__device__ __forceinline__ void wg_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
}

typedef double d4 __attribute__((ext_vector_type(4)));
// explicit global address space: a generic pointer that went through an opaque asm loads with flat_load,
// which also counts on lgkmcnt and would make every LDS-operand wait an L2 round trip
typedef const double __attribute__((address_space(1)))* gptr_t;


// ------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  key = (seed_lo, seed_hi), counter = (row_lo, row_hi,
// sweep_lo, sweep_hi); the uniform is the 53-bit float built from the first two output words.
// ------------------------------------------------------------------------------------------
__host__ __device__ inline double philox_uniform(uint64_t seed, uint64_t row, uint64_t sweep) {
  uint32_t c0 = (uint32_t)row, c1 = (uint32_t)(row >> 32);
  uint32_t c2 = (uint32_t)sweep, c3 = (uint32_t)(sweep >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return ((double)(c0 >> 5) * 67108864.0 + (double)(c1 >> 6)) * (1.0 / 9007199254740992.0);
}

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}


// exp(x) for x <= 0 in 10 float64 pipe operations + 4 integer ones (the f64 VALU shares its pipe with the
// f64 MFMA on gfx950 and the normalise phase is bound by its instruction count, so every instruction of
// the softmax is paid in matrix issue slots):
//   n = rint(x * 64/ln2) via the 1.5*2^52 trick, r = x - n ln2/64 in ONE fma (|r| <= ln2/128),
//   exp(x) = 2^(n>>6) * tab[n & 63] * (1 + r + ... + r^5/120),  tab[j] = 2^(j/64) in LDS.
// The single-constant reduction leaves an error of |n| * 1.2e-18 in r, i.e. a relative error of
// 1.1e-16 * |x| in the result — an ABSOLUTE error below 4e-17 for every x <= 0 (max of |x| e^x), which is
// what a sum of exponentials whose largest term is 1 sees.  The argument is clamped at -707 (also -inf and the
// -1e300 of padding components): everything below returns exp(-707) = 8e-308, the smallest value whose exponent
// field the integer add below cannot underflow — zero for every purpose here (sums whose largest term is 1), and
// one v_max_f64 instead of a 64-bit compare and two selects per element.  The shift by 6 goes through an opaque
// asm: LLVM otherwise rewrites ((n >> 6) << 20) + hi as shift, mask and a 64-bit add (3 instructions for 2).
__device__ inline double exp_nonpos(double x, const double* __restrict__ tab) {
  x = fmax(x, -707.0);
  const double t = fma(x, 92.33248261689366, 6755399441055744.0);
  const int n = __double2loint(t);
  const double nf = t - 6755399441055744.0;
  const double r = fma(nf, -0x1.62e42fefa39efp-7, x);
  double q = fma(r, 1.0 / 120.0, 1.0 / 24.0);
  q = fma(r, q, 1.0 / 6.0);
  q = fma(r, q, 0.5);
  q = fma(r, q, 1.0);
  const double e = tab[n & 63] * fma(r, q, 1.0);
  int n6;
  asm("v_ashrrev_i32 %0, 6, %1" : "=v"(n6) : "v"(n));
  return __hiloint2double(__double2hiint(e) + (n6 << 20), __double2loint(e));
}

// exp(x), x <= 0, with a 2048-entry table of 2^(i/2048) (16 KB of LDS, where a kernel has them to spare: the row-owner kernels, the small-shape kernel): the reduced
// argument is <= ln2/4096, a cubic is enough (r^4/24 < 4e-17) — 13 instructions instead of the 15 of exp_nonpos, and
// the row-owner label kernel issues 64 of them per lane and step.  Same argument clamp, same single-constant reduction (relative
// error 1.1e-16 |x|, an absolute error below 4e-17 for every x <= 0).
constexpr int kExpTab = 2048;
__device__ __forceinline__ double exp_nonpos_t2048(double x, const double* __restrict__ tab) {
  x = fmax(x, -707.0);
  const double t = fma(x, 2954.639443740597, 6755399441055744.0);          // 2048 / ln2, 1.5 * 2^52
  const int n = __double2loint(t);
  const double nf = t - 6755399441055744.0;
  const double r = fma(nf, -3.3845077175778103e-04, x);                    // ln2 / 2048
  double q = fma(r, 1.0 / 6.0, 0.5);
  q = fma(r, q, 1.0);
  const double e = tab[n & (kExpTab - 1)] * fma(r, q, 1.0);
  int nh;
  asm("v_ashrrev_i32 %0, 11, %1" : "=v"(nh) : "v"(n));
  return __hiloint2double(__double2hiint(e) + (nh << 20), __double2loint(e));
}

// The same exponential with ONE integer instruction for the factor 2^(n >> 11) instead of two: table entry i holds the bit
// pattern of 2^(i/2048) with (i << 9) subtracted from its high word, so that adding (n << 9) — n = 2048 (n >> 11) + i — to
// the high word lands on hi(2^(i/2048)) + ((n >> 11) << 20): the low bits of n cancel by construction (v_lshl_add_u32 in
// place of v_ashrrev_i32 + v_lshl_add_u32).  12 instructions with the subtraction of the row maximum; same error bound.
__device__ __forceinline__ double exp_tab_entry_c(int i) {
  const double v = exp2((double)i * (1.0 / kExpTab));
  return __hiloint2double(__double2hiint(v) - (i << 9), __double2loint(v));
}
__device__ __forceinline__ double exp_nonpos_t2048c(double x, const double* __restrict__ tab) {
  x = fmax(x, -707.0);
  const double t = fma(x, 2954.639443740597, 6755399441055744.0);          // 2048 / ln2, 1.5 * 2^52
  const int n = __double2loint(t);
  const double nf = t - 6755399441055744.0;
  const double r = fma(nf, -3.3845077175778103e-04, x);                    // ln2 / 2048
  double q = fma(r, 1.0 / 6.0, 0.5);
  q = fma(r, q, 1.0);
  const double tv = tab[n & (kExpTab - 1)];
  int hi;
  asm("v_lshl_add_u32 %0, %1, 9, %2" : "=v"(hi) : "v"(n), "v"(__double2hiint(tv)));
  return __hiloint2double(hi, __double2loint(tv)) * fma(r, q, 1.0);
}

// ... in two halves, so that a kernel can put a batch of table reads in flight before it needs the first of them (hipcc
// otherwise keeps two lookups in flight and waits ~25 cycles after each read: the LDS round trip of every pair of
// exponentials lay open in narrow_kernel, 31 % of the pipe idle at three waves per SIMD):
//   exp_c_issue:  reduced argument r and the integer n (table index n & 2047, scale n >> 11);   the caller reads tab[n & 2047]
//   exp_c_finish: cubic in r times the scaled table entry
__device__ __forceinline__ void exp_c_issue(double x, double& r, int& n) {
  x = fmax(x, -707.0);
  const double t = fma(x, 2954.639443740597, 6755399441055744.0);
  n = __double2loint(t);
  r = fma(t - 6755399441055744.0, -3.3845077175778103e-04, x);
}
__device__ __forceinline__ double exp_c_finish(double r, int n, double tv) {
  double q = fma(r, 1.0 / 6.0, 0.5);
  q = fma(r, q, 1.0);
  int hi;
  asm("v_lshl_add_u32 %0, %1, 9, %2" : "=v"(hi) : "v"(n), "v"(__double2hiint(tv)));
  return __hiloint2double(hi, __double2loint(tv)) * fma(r, q, 1.0);
}

// ------------------------------------------------------------------------------------------
// Short dependency chains.  The f64 VALU shares its pipe with the f64 MFMA, and the wave of the OTHER
// workgroup on this SIMD is usually inside a matrix phase: every time this wave has no ready f64
// instruction (it waits for the previous result), the pipe goes to an MFMA for 64 cycles.  A dependent
// chain therefore costs ~70 cycles per link, an independent group ~6 per instruction (measured with
// what-if builds at C3: the 32-link max / running-sum / compare chains were 45 % of the normalise phase).
// All reductions below are trees, the cumulative sum is a Kogge-Stone scan.
// ------------------------------------------------------------------------------------------
// (mimo_kernels.hip is compiled with -fno-honor-nans: without it LLVM puts a canonicalising self-max in front
// of every fmax whose operands may be signalling NaNs — anything that came out of memory or a lane exchange —
// which doubled the f64 instructions of the max tree: 21 -> 10 v_max_f64 per lane at K = 64.  No kernel here
// computes with NaNs: the host rejects NaN input.)
__device__ __forceinline__ double tree_max8(const double (&v)[8]) {
  return fmax(fmax(fmax(v[0], v[1]), fmax(v[2], v[3])), fmax(fmax(v[4], v[5]), fmax(v[6], v[7])));
}
__device__ __forceinline__ double tree_sum8(const double (&v)[8]) {   // == scan8(v)[7], bit for bit
  return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
}
__device__ __forceinline__ void scan8(double (&x)[8]) {               // inclusive, 3 levels
  double a[8], b[8];
  a[0] = x[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) a[i] = x[i - 1] + x[i];
  b[0] = a[0]; b[1] = a[1];
#pragma unroll
  for (int i = 2; i < 8; ++i) b[i] = a[i - 2] + a[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = b[i];
#pragma unroll
  for (int i = 4; i < 8; ++i) x[i] = b[i - 4] + b[i];
}

}  // namespace mimo
