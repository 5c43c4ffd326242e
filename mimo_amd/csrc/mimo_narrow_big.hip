// Instantiations of the narrow kernels (mimo_narrow_kernel.h) for 129 .. 256 components over one or two contraction steps
// (at most 8 features: the full map of Dz <= 2 — the reference's ILR examples with their --nb_models ceiling raised,
// examples/ilr/evaluate_sinc.py:35 — and reduced maps of that size): V = 36 .. 64 slots per lane, one per 16-component band
// of the partial block.  A translation unit of its own (compiles beside the others).
#include "mimo_narrow_kernel.h"

namespace mimo {

template <int V>
static narrow_fn pick_narrow_big_nsf(int nsf, int gibbs, int zi) {
  switch (nsf) {
    case 1: return pick_narrow_mode<V, 1>(gibbs, zi);
    case 2: return pick_narrow_mode<V, 2>(gibbs, zi);
    // three / four steps (Dz = 3, 4): the label pass only — it has no accumulators; V NSF accumulators of the softmax pass do not fit
    case 3: return gibbs == 1 ? narrow_kernel<V, 3, 1, 1> : nullptr;
    case 4: return gibbs == 1 ? narrow_kernel<V, 4, 1, 1> : nullptr;
  }
  return nullptr;
}
narrow_fn pick_narrow_big(int V, int nsf, int gibbs, int zi) {
  if (zi != 1) return nullptr;
  switch (V) {
#define MIMO_NV(v) case v: return pick_narrow_big_nsf<v>(nsf, gibbs, zi);
    MIMO_NV(36) MIMO_NV(40) MIMO_NV(44) MIMO_NV(48) MIMO_NV(52) MIMO_NV(56) MIMO_NV(60) MIMO_NV(64)
#undef MIMO_NV
  }
  return nullptr;
}

}  // namespace mimo
