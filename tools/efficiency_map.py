"""Diagnostic: fraction of the FP64 peak (78.6 TFLOP/s, algorithmic flops of SURVEY.md section 8(d)) of one VI pass and one
Gibbs sweep over a grid of (Dz, K), device time of all kernels of the pass (HIP events).
    python tools/efficiency_map.py [N]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd.engine import HipEngine
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
Ds = (2, 4, 8, 12, 16, 20, 24, 28, 32)
Ks = (4, 8, 16, 32, 48, 64, 96, 128, 192, 256)
eng = HipEngine(0)
rng = np.random.default_rng(0)
for mode in ("vi", "gibbs"):
    print(f"{mode}: fraction of 78.6 TFLOP/s (N = {N}); rows Dz, columns K = {Ks}")
    for D in Ds:
        Z = rng.standard_normal((N, D)); eng.upload(Z)
        line = []
        for K in Ks:
            A = rng.standard_normal((K, D, D)); W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D)
            b = rng.standard_normal((K, D)); c = rng.standard_normal(K)
            run = (lambda it: eng.gibbs_labels(c, b, W, seed=1, sweep=it, return_labels=False)) if mode == "gibbs" else (lambda it: eng.estep(c, b, W))
            for it in range(2): run(it)
            eng.profile(True); eng.profile_read(reset=True)
            for it in range(4): run(it)
            ms, n = eng.profile_read(reset=True); eng.profile(False)
            FE, FS = D * (D + 1) + 3 * D + 8, (D + 1) * (D + 2) + 1
            flops = N * K * FE + (N * K * FS if mode == "vi" else N * FS)
            line.append(flops / (ms / n * 1e-3) / 78.6e12)
        print(f"  Dz={D:2d} " + " ".join(f"{v:5.2f}" for v in line), flush=True)
