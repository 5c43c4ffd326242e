"""Diagnostic: Gibbs sweep (label pass + label statistics) of the mid shapes on the row-owner label kernels and on the mid kernel's label
mode (mimo_tune "mid_labels_min_d"), one box.  ms per sweep and fraction of 78.6 TFLOP/s (N K F_E + N F_S flops).
    python tools/mid_label_sweep.py [N] ["D,K D,K ..."]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd.engine import HipEngine
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
shapes = [tuple(map(int, s.split(","))) for s in sys.argv[2].split()] if len(sys.argv) > 2 else \
    [(D, K) for D in (10, 12, 14, 16, 17, 20, 24, 28, 32) for K in (8, 16, 24, 32, 48)]
eng = HipEngine(0)
rng = np.random.default_rng(0)
last_D = None
print(f"N = {N}; per shape: default route | mid label mode: kind, ms per sweep (fraction of 78.6 TFLOP/s)")
for D, K in shapes:
    if D != last_D:
        Z = rng.standard_normal((N, D)); eng.upload(Z); last_D = D
    A = rng.standard_normal((K, D, D)); W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D)
    b = rng.standard_normal((K, D)); c = rng.standard_normal(K)
    FE, FS = D * (D + 1) + 3 * D + 8, (D + 1) * (D + 2) + 1
    flops = N * K * FE + N * FS
    out = []
    for mind in (64, 10):
        eng.tune("mid_labels_min_d", mind); eng.tune("mid_labels_narrow_k", int(os.environ.get("MID_LABELS_NARROW_K", "0")))
        kind = eng.plan(K, gibbs=True)["kind"]
        for it in range(2): eng.gibbs_labels(c, b, W, seed=1, sweep=it, return_labels=False)
        eng.profile(True); eng.profile_read(reset=True)
        for it in range(4): eng.gibbs_labels(c, b, W, seed=1, sweep=it, return_labels=False)
        ms, n = eng.profile_read(reset=True); eng.profile(False)
        out.append((kind, ms / n, flops / (ms / n * 1e-3) / 78.6e12))
    eng.tune("mid_labels_min_d", 0); eng.tune("mid_labels_narrow_k", 0)
    print(f"Dz={D:2d} K={K:3d}: " + "  |  ".join(f"{o[0]:8s} {o[1]:7.3f} ms ({o[2]:.2f})" for o in out), flush=True)
