"""Diagnostic: wall time of the statistics-only passes (device-resident table / labels) and of the two-stage sweep.
python tools/quick_stats_time.py N D K"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd.engine import HipEngine
N = int(float(sys.argv[1])); D = int(sys.argv[2]); K = int(sys.argv[3])
rng = np.random.default_rng(0)
Z = rng.standard_normal((N, D)); A = rng.standard_normal((K, D, D))
W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D); b = rng.standard_normal((K, D)); c = rng.standard_normal(K)
eng = HipEngine(0); eng.upload(Z)
eng.estep(c, b, W, keep_resp=True)
eng.gibbs_labels(c, b, W, seed=1, sweep=0, stats=False, return_labels=False)
def med(f):
    for _ in range(3): f()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3
print(f"N={N} D={D} K={K} split_stats={os.environ.get('MIMO_SPLIT_STATS', '1')}: weighted_stats(resident) {med(lambda: eng.weighted_stats(None, K)):.3f} ms, "
      f"label_stats(resident) {med(lambda: eng.label_stats(None, K)):.3f} ms, estep {med(lambda: eng.estep(c, b, W)):.3f} ms")
