import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
from mimo_amd.engine import HipEngine
eng = HipEngine(0)
N = 4_000_000
for D, K in ((32, 128), (20, 64)):
    rng = np.random.default_rng(0)
    Z = rng.standard_normal((N, D)); eng.upload(Z)
    lab = rng.integers(0, K, size=N).astype(np.int32)
    for it in range(6): eng.label_stats(lab, K)
