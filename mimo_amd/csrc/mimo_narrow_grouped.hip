// Instantiations of the grouped (Dz-templated) narrow kernels (mimo_narrow_kernel.h): full feature map, Dz = 5 .. 32.
#include "mimo_narrow_kernel.h"

namespace mimo {

// ---- the grouped (Dz-templated) variant: full feature map, Dz = 5 .. 32 ------------------------------------------------
// instantiated while the V x steps accumulators fit the unified register file of one wave per SIMD; the label pass for the
// shapes whose labels have a label-statistics kernel behind them (Dz <= 16)
constexpr int kNarrowGroupMaxAcc = 200;
template <int D, int V>
static narrow_fn pick_narrow_dt_mode(int gibbs) {
  constexpr int NST = narrow_group_steps(D), ZI = D <= 16 ? 4 : 8;
  if (gibbs == 1) {
    if constexpr (D <= 16) return narrow_kernel<V, NST, 1, ZI, D>;
  } else {
    if constexpr (V * NST <= kNarrowGroupMaxAcc) return gibbs == 2 ? narrow_kernel<V, NST, 2, ZI, D> : narrow_kernel<V, NST, 0, ZI, D>;
  }
  return nullptr;
}
template <int D>
static narrow_fn pick_narrow_dt_v(int V, int gibbs) {
  switch (V) {
    case 1: return pick_narrow_dt_mode<D, 1>(gibbs);
    case 2: return pick_narrow_dt_mode<D, 2>(gibbs);
    case 3: if constexpr (D <= 16) return pick_narrow_dt_mode<D, 3>(gibbs); else return nullptr;
    case 4: if constexpr (D <= 16) return pick_narrow_dt_mode<D, 4>(gibbs); else return nullptr;
    case 6: if constexpr (D <= 8) return pick_narrow_dt_mode<D, 6>(gibbs); else return nullptr;
  }
  return nullptr;
}
narrow_fn pick_narrow_dt(int V, int D, int gibbs) {
  switch (D) {
#define MIMO_ND(d) case d: return pick_narrow_dt_v<d>(V, gibbs);
    MIMO_ND(5) MIMO_ND(6) MIMO_ND(7) MIMO_ND(8) MIMO_ND(9) MIMO_ND(10) MIMO_ND(11) MIMO_ND(12) MIMO_ND(13) MIMO_ND(14) MIMO_ND(15) MIMO_ND(16)
    MIMO_ND(17) MIMO_ND(18) MIMO_ND(19) MIMO_ND(20) MIMO_ND(21) MIMO_ND(22) MIMO_ND(23) MIMO_ND(24) MIMO_ND(25) MIMO_ND(26) MIMO_ND(27)
    MIMO_ND(28) MIMO_ND(29) MIMO_ND(30) MIMO_ND(31) MIMO_ND(32)
#undef MIMO_ND
  }
  return nullptr;
}

}  // namespace mimo
