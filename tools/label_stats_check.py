"""Diagnostic: mimo_label_stats of a caller's label vector at Dz > 16 (one-pass kernel over the ranked tiles), small and large K, against the oracle."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from mimo_amd.engine import HipEngine
from oracle import mimo_oracle as O
eng = HipEngine(0)
rng = np.random.default_rng(3)
bad = 0
for (N, D, K) in [(30011, 20, 6), (70001, 32, 12), (5, 17, 3), (140003, 24, 16), (1, 32, 1), (257, 28, 300 % 256), (99999, 19, 200)]:
    Z = rng.standard_normal((N, D)); lab = rng.integers(0, K, size=N).astype(np.int32)
    eng.upload(Z)
    S = eng.label_stats(lab, K)
    ok = (lab >= 0) & (lab < K)
    R = np.zeros((K, N)); R[lab[ok], np.nonzero(ok)[0]] = 1.
    n, sx, sxx = O.packed_stats(Z, R)
    e = max(np.abs(S.n - n).max(), np.abs(S.sx - sx).max() / max(1, np.abs(sx).max()), np.abs(S.sxx - sxx).max() / max(1, np.abs(sxx).max()))
    S2 = eng.label_stats(lab, K)
    same = np.array_equal(S2.sxx, S.sxx)
    print(N, D, K, "err %.1e" % e, "same", same)
    bad += (e > 1e-11) or not same
print("bad", bad)
