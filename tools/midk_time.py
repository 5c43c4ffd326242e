"""Diagnostic: Gibbs sweep (labels + statistics) for mid-size K on the tile kernels vs the row-owner kernels (MIMO_ROWWAVE_MIN_K)."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(16, 64), (16, 32), (16, 16), (12, 64), (12, 128), (10, 64), (13, 40), (16, 4)]
if len(sys.argv) > 2 and sys.argv[2] == "child":
    from mimo_amd.engine import HipEngine
    N = int(float(sys.argv[1])); eng = HipEngine(0)
    for D, K in SHAPES:
        rng = np.random.default_rng(0)
        Z = rng.standard_normal((N, D)); A = rng.standard_normal((K, D, D))
        W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D); b = rng.standard_normal((K, D)); c = rng.standard_normal(K)
        eng.upload(Z)
        run = lambda it: eng.gibbs_labels(c, b, W, seed=1, sweep=it, return_labels=False)
        for it in range(3): run(it)
        eng.profile(True); eng.profile_read(reset=True)
        for it in range(8): run(it)
        ms, n = eng.profile_read(reset=True); eng.profile(False)
        print(f"D={D} K={K:3d} {eng.plan(K, gibbs=True)['kind']:8s} sweep kernels {ms / n:.3f} ms", flush=True)
else:
    N = sys.argv[1] if len(sys.argv) > 1 else "1e7"
    for mk in ("300", "1"):
        print(f"--- MIMO_ROWWAVE_MIN_K={mk}", flush=True)
        subprocess.run([sys.executable, __file__, N, "child"], env=dict(os.environ, MIMO_ROWWAVE_MIN_K=mk), check=True)
