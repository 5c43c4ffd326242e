"""Diagnostic: per-kernel device time (HIP events, mimo_profile_kernels) of one VI pass and one Gibbs sweep.
    python tools/kernel_breakdown.py "D,K D,K ..." [N]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd.engine import HipEngine
N = int(float(sys.argv[2])) if len(sys.argv) > 2 else 2_000_000
eng = HipEngine(0)
for sh in sys.argv[1].split():
    D, K = map(int, sh.split(","))
    rng = np.random.default_rng(0)
    Z = rng.standard_normal((N, D)); A = rng.standard_normal((K, D, D))
    W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D); b = rng.standard_normal((K, D)); c = rng.standard_normal(K)
    eng.upload(Z)
    for gibbs in (False, True):
        run = (lambda it: eng.gibbs_labels(c, b, W, seed=1, sweep=it, return_labels=False)) if gibbs else (lambda it: eng.estep(c, b, W))
        for it in range(2): run(it)
        eng.profile(True); eng.profile_read(reset=True)
        for it in range(5): run(it)
        kern = eng.profile_kernels(); ms, n = eng.profile_read(reset=True); eng.profile(False)
        F = (D + 1) * (D + 2) // 2
        print(f"D={D} K={K} N={N} {'gibbs' if gibbs else 'vi'}: {ms / n:.3f} ms per pass")
        for name, v in kern.items():
            per = v["ms"] / v["launches"]
            print(f"    {name:48s} {v['launches'] // 5} x {per:8.3f} ms")
