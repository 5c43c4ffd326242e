"""Diagnostic: softmax + statistics pass on the tile kernels vs the row-owner kernel (MIMO_ROWWAVE_VI) for K <= 64, Dz <= 9."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(8, 64), (8, 48), (8, 32), (8, 24), (8, 16), (8, 8), (8, 4), (5, 64), (5, 32), (5, 16), (5, 7), (7, 33), (9, 32),
          (9, 16), (6, 40), (3, 64), (3, 24), (2, 50), (4, 20), (1, 64)]
if len(sys.argv) > 2 and sys.argv[2] == "child":
    from mimo_amd.engine import HipEngine
    N = int(float(sys.argv[1])); eng = HipEngine(0)
    for D, K in SHAPES:
        rng = np.random.default_rng(0)
        Z = rng.standard_normal((N, D)); A = rng.standard_normal((K, D, D))
        W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D); b = rng.standard_normal((K, D)); c = rng.standard_normal(K)
        eng.upload(Z)
        run = lambda it: eng.estep(c, b, W)
        for it in range(3): run(it)
        eng.profile(True); eng.profile_read(reset=True)
        for it in range(8): run(it)
        ms, n = eng.profile_read(reset=True); eng.profile(False)
        print(f"D={D} K={K:3d} {eng.plan(K)['kind']:10s} pass kernels {ms / n:.3f} ms", flush=True)
else:
    N = sys.argv[1] if len(sys.argv) > 1 else "1e7"
    for v in ("0", "1"):
        print(f"--- MIMO_ROWWAVE_VI={v}", flush=True)
        subprocess.run([sys.executable, __file__, N, "child"], env=dict(os.environ, MIMO_ROWWAVE_VI=v), check=True)
