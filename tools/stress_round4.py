"""Stress: the round-4 kernels — mid kernels (one to eight row blocks, one wave per SIMD variants, label mode), Gram label statistics,
narrow kernels for 129 .. 256 components, the bound-only pass on the fast kernels — launched many times on several shapes: every launch
must return the bits of the first one and agree with the oracle (labels exact).
    python tools/stress_round4.py [launches]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd.engine import HipEngine
from oracle import mimo_oracle as O
from scipy.special import logsumexp
R = int(sys.argv[1]) if len(sys.argv) > 1 else 150
eng = HipEngine(0)
bad = 0
for (N, D, K) in ((200003, 20, 16), (150011, 24, 32), (120007, 32, 16), (100003, 28, 48), (150011, 16, 96), (120007, 20, 80),    # mid, 1 - 6 row blocks
                  (100003, 12, 112), (90001, 14, 128), (100003, 26, 80), (80021, 32, 48),                                   # mid: seven / eight row blocks, one wave per SIMD
                  (200003, 17, 9), (150011, 24, 8), (180001, 12, 40),                                                       # label mode incl. K = 5 .. 8 at Dz >= 24
                  (400003, 2, 160), (300007, 2, 192), (300007, 1, 256), (250013, 2, 256), (250013, 3, 200)):               # big narrow (softmax / label pass)
    rng = np.random.default_rng(N % 1000 + D + K)
    Z = rng.standard_normal((N, D)) * 1.5; A = rng.standard_normal((K, D, D))
    W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D); mu = rng.standard_normal((K, D)) * 2
    b = np.einsum('kde,ke->kd', W, mu); c = -0.5 * np.einsum('kd,kd->k', mu, b)
    eng.upload(Z)
    L = O.canonical_eval(Z, c, b, W)
    ref = O.sample_discrete_from_log(L, O.philox_uniforms(7, np.arange(N), 2))
    lse = logsumexp(L, axis=0)
    n0, _, sxx0 = O.packed_stats(Z, np.exp(L - lse))
    first = None
    for r in range(R):
        S, sc = eng.estep(c, b, W)
        lab, G = eng.gibbs_labels(c, b, W, seed=7, sweep=2)
        _, scb = eng.estep(c, b, W, stats=False)               # the bound-only pass (promoted to the plain pass on narrow / small mid shapes)
        cur = (S.sxx.tobytes(), S.n.tobytes(), sc[0], G.sxx.tobytes(), G.n.tobytes(), scb[0])
        if first is None:
            first = cur
            assert np.abs(S.sxx - sxx0).max() / np.abs(sxx0).max() < 1e-11 and np.abs(S.n - n0).max() / n0.max() < 1e-11
            assert abs(sc[0] - lse.sum()) < 1e-12 * abs(lse.sum())
        if cur != first or not np.array_equal(lab, ref):
            bad += 1
            print(f"MISMATCH N={N} D={D} K={K} launch {r}: labels differ {int((lab != ref).sum())}", flush=True)
    print(f"N={N} D={D} K={K} plan vi={eng.plan(K)['kind']} gibbs={eng.plan(K, gibbs=True)['kind']}: {R} launches ok", flush=True)
print("bad launches:", bad)
sys.exit(1 if bad else 0)
