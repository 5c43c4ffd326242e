"""Stress: the round-2 kernels (small-shape kernel, row-owner label kernel, label statistics, wide E-step / statistics) launched many times on
several shapes — every launch must return the bits of the first one and agree with the oracle (labels exact).
    python tools/stress_round2.py [launches]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd.engine import HipEngine
from oracle import mimo_oracle as O
from scipy.special import logsumexp
R = int(sys.argv[1]) if len(sys.argv) > 1 else 150
eng = HipEngine(0)
bad = 0
for (N, D, K) in ((3 * 1024 * 256 + 5, 2, 4), (400003, 2, 25), (300007, 4, 16), (3 * 8 * 256 * 16 + 9, 8, 256), (500009, 8, 64),
                  (400001, 5, 16), (350003, 9, 130), (200003, 1, 32), (300011, 16, 64), (250007, 12, 128), (200009, 13, 20),
                  # two-stage shapes on the wide kernels (mimo_wide.hip)
                  (150011, 32, 128), (120007, 16, 200), (130003, 24, 100), (100003, 20, 72)):
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
        cur = (S.sxx.tobytes(), S.n.tobytes(), sc[0], G.sxx.tobytes(), G.n.tobytes())
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
