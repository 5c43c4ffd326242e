"""Repeated launches at shapes that exercise every kernel family with two workgroups per CU: every run must be
bit-identical to the first and the first must match the oracle.  (Found the generic-mode failure described at
MIMO_GENERIC_1WG_NCB in mimo_kernels.hip.)   python tools/stress_generic.py [reps]"""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from mimo_amd.engine import HipEngine
from oracle import mimo_oracle as O
from scipy.special import logsumexp
from test_gpu_parity import _random_problem
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
eng = HipEngine(0)
tot = 0
shapes = [(15, 64), (16, 64), (8, 256), (12, 100), (9, 256), (32, 128)]
if len(sys.argv) > 2:
    shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[2:]]
for D, K in shapes:
    N = 32 * 512 * 3 + 77
    rng = np.random.default_rng(100 * D + K)
    Z, c, b, W = _random_problem(rng, N, D, K)
    eng.upload(Z)
    L = O.canonical_eval(Z, c, b, W); lse = logsumexp(L, axis=0)
    n, sx, sxx = O.packed_stats(Z, np.exp(L - lse))
    for name, kw in [("fast", {}), ("keep_lse", dict(keep_lse=True)), ("split", dict(entropy_split=True))]:
        bad = 0
        for r in range(reps):
            S, sc = eng.estep(c, b, W, **kw)
            bad += np.abs(S.sxx - sxx).max() > 1e-9 * np.abs(sxx).max()
        print(f"D={D} K={K} VI {name}: {bad} bad of {reps}"); tot += bad
    ref = O.sample_discrete_from_log(L, O.philox_uniforms(5, np.arange(N), 1))
    bad = 0
    for r in range(reps):
        labels, S = eng.gibbs_labels(c, b, W, seed=5, sweep=1)
        bad += (not np.array_equal(labels, ref)) or abs(S.n.sum() - N) > 1e-6
    print(f"D={D} K={K} Gibbs: {bad} bad of {reps}"); tot += bad
print("TOTAL BAD", tot)
