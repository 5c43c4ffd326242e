"""Stress: 150 launches of a VI pass and a Gibbs sweep on six two-stage / large-K shapes, every launch bit-identical to the first."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd.engine import HipEngine
eng = HipEngine(0); bad = 0
for (N, D, K) in ((600011, 32, 128), (500009, 16, 200), (450007, 24, 144), (400003, 12, 160), (350003, 28, 40), (500003, 8, 192)):
    rng = np.random.default_rng(N % 1000 + D + K)
    Z = rng.standard_normal((N, D)) * 1.5; A = rng.standard_normal((K, D, D))
    W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D); mu = rng.standard_normal((K, D)) * 2
    b = np.einsum('kde,ke->kd', W, mu); c = -0.5 * np.einsum('kd,kd->k', mu, b)
    eng.upload(Z); first = None
    for r in range(150):
        S, sc = eng.estep(c, b, W)
        lab, G = eng.gibbs_labels(c, b, W, seed=7, sweep=2)
        cur = (S.sxx.tobytes(), S.n.tobytes(), sc[0], G.sxx.tobytes(), G.n.tobytes(), lab.tobytes())
        if first is None: first = cur
        elif cur != first:
            bad += 1; print("MISMATCH", N, D, K, r, flush=True)
    print(f"N={N} D={D} K={K}: 150 launches identical", flush=True)
print("bad", bad); sys.exit(1 if bad else 0)
