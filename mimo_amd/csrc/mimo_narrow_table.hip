// Instantiations of the table-driven narrow kernels (mimo_narrow_kernel.h): five contraction steps and more (17+ features).
#include "mimo_narrow_kernel.h"

namespace mimo {

template <int V>
static narrow_fn pick_narrow_table_nsf(int nsf, int gibbs, int zi) {
  switch (nsf) {
#define MIMO_NN(n) case n: return pick_narrow_mode<V, n>(gibbs, zi);
    // ceil(F / 4) of the full maps of Dz = 5 .. 16 (F = 21 .. 153) and of the reduced maps beyond 16 features
    MIMO_NN(5) MIMO_NN(6) MIMO_NN(7) MIMO_NN(8) MIMO_NN(9) MIMO_NN(12) MIMO_NN(14) MIMO_NN(17) MIMO_NN(20) MIMO_NN(23)
    MIMO_NN(27) MIMO_NN(30) MIMO_NN(34) MIMO_NN(39)
    // Dz = 17 .. 32 (F = 171 .. 561)
    MIMO_NN(43) MIMO_NN(48) MIMO_NN(53) MIMO_NN(58) MIMO_NN(64) MIMO_NN(69) MIMO_NN(75) MIMO_NN(82) MIMO_NN(88) MIMO_NN(95)
    MIMO_NN(102) MIMO_NN(109) MIMO_NN(117) MIMO_NN(124) MIMO_NN(132) MIMO_NN(141)
#undef MIMO_NN
  }
  return nullptr;
}
narrow_fn pick_narrow_table(int V, int nsf, int gibbs, int zi) {
  switch (V) {
#define MIMO_NV(v) case v: return pick_narrow_table_nsf<v>(nsf, gibbs, zi);
    MIMO_NV(1) MIMO_NV(2) MIMO_NV(3) MIMO_NV(4) MIMO_NV(6) MIMO_NV(8) MIMO_NV(10) MIMO_NV(12) MIMO_NV(13) MIMO_NV(14) MIMO_NV(16)
#undef MIMO_NV
  }
  return nullptr;
}

}  // namespace mimo
