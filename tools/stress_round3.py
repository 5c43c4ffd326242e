"""Stress: the round-3 kernels — narrow kernels (slot loops of Dz <= 4, table-driven and grouped loops, label draw + statistics in one pass),
streamed row-owner label kernel (partial last chunk), slot-table label statistics with the histogram counted by the label kernel, fused
reduce + unpack — launched many times on several shapes: every launch must return the bits of the first one and agree with the oracle
(labels exact).
    python tools/stress_round3.py [launches]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd.engine import HipEngine
from oracle import mimo_oracle as O
from scipy.special import logsumexp
R = int(sys.argv[1]) if len(sys.argv) > 1 else 150
eng = HipEngine(0)
bad = 0
for (N, D, K) in ((400003, 2, 50), (300007, 1, 100), (250013, 4, 128), (300011, 3, 33),             # narrow, Dz <= 4
                  (300007, 8, 4), (250007, 16, 4), (200003, 12, 13), (260003, 5, 24), (200003, 16, 16),   # table-driven / grouped loops
                  (150011, 24, 8), (120007, 32, 4), (130003, 20, 3), (140009, 28, 2),                      # Dz > 16: one wave per SIMD
                  (3 * 8 * 256 * 16 + 9, 8, 256), (600011, 9, 200), (500009, 6, 40),                        # slot-table label statistics
                  (150011, 16, 128), (130003, 20, 64), (120007, 12, 256), (100003, 24, 96), (90001, 32, 128),    # streamed label kernel
                  (700001, 20, 12), (650003, 32, 200), (1200007, 17, 40)):            # one-pass label statistics over several ranges per workgroup
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
