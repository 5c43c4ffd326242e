"""Diagnostic: device time (HIP events, all kernels of the pass) of one VI pass and one Gibbs sweep at the small-Dz, mid-K shapes the
reference's own ILR examples default to (examples/ilr/evaluate_sine.py: dx = dy = 1, 50 experts; evaluate_sinc.py: 100), and the
kernel family mimo_plan routes each to.
    python tools/narrow_sweep.py [N]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd.engine import HipEngine
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
Ds = (1, 2, 3, 4)
Ks = (33, 48, 50, 64, 96, 100, 128)
eng = HipEngine(0)
rng = np.random.default_rng(0)
for mode in ("vi", "gibbs"):
    print(f"{mode}: ms per pass / fraction of 78.6 TFLOP/s / route (N = {N}); rows Dz, columns K = {Ks}")
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
            kind = eng.plan(K, gibbs=(mode == "gibbs")).get("kind", "?")
            line.append(f"{ms / n:6.3f}/{flops / (ms / n * 1e-3) / 78.6e12:4.2f}/{kind[:5]}")
        print(f"  Dz={D:2d} " + " ".join(line), flush=True)
