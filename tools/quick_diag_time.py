"""Diagnostic: fused sweep under the diagonal structure (2 Dz + 1 features).  python tools/quick_diag_time.py N D K [gibbs]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd.engine import HipEngine
N = int(float(sys.argv[1])); D = int(sys.argv[2]); K = int(sys.argv[3])
GIBBS = len(sys.argv) > 4 and sys.argv[4] == "gibbs"
rng = np.random.default_rng(0)
Z = rng.standard_normal((N, D))
W = np.zeros((K, D, D)); W[:, np.arange(D), np.arange(D)] = rng.uniform(0.5, 2., (K, D))
b = rng.standard_normal((K, D)); c = rng.standard_normal(K)
eng = HipEngine(0); eng.set_structure('diag'); eng.upload(Z)
def run(it):
    if GIBBS: eng.gibbs_labels(c, b, W, seed=1, sweep=it, return_labels=False)
    else: eng.estep(c, b, W)
for it in range(3): run(it)
ts = []
for it in range(10):
    t0 = time.perf_counter(); run(it); ts.append(time.perf_counter() - t0)
print(f"diag N={N} D={D} K={K} {'gibbs' if GIBBS else 'vi'} split_table={os.environ.get('MIMO_SPLIT_TABLE', '1')}: {float(np.median(ts))*1e3:.3f} ms")
