"""Diagnostic: the Gibbs sweep as ONE fused launch (labels + statistics) against the unfused pair
(label-only launch, then the label-indexed statistics pass).  python tools/c3_unfused_time.py N D K"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd.engine import HipEngine
N = int(float(sys.argv[1])); D = int(sys.argv[2]); K = int(sys.argv[3])
rng = np.random.default_rng(0)
Z = rng.standard_normal((N, D)); A = rng.standard_normal((K, D, D))
W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D); b = rng.standard_normal((K, D)); c = rng.standard_normal(K)
eng = HipEngine(0); eng.upload(Z)


def timed(fn, reps=8):
    for it in range(2): fn(it)
    ts = []
    for it in range(reps):
        t0 = time.perf_counter(); fn(it); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


def kernel_ms(fn, reps=6):
    for it in range(2): fn(it)
    eng.profile(True); eng.profile_read(reset=True)
    for it in range(reps): fn(it)
    ms, n = eng.profile_read(reset=True)
    eng.profile(False)
    return ms / max(n, 1), n / reps


fused = lambda it: eng.gibbs_labels(c, b, W, seed=1, sweep=it, return_labels=False)
lab = lambda it: eng.gibbs_labels(c, b, W, seed=1, sweep=it, stats=False, return_labels=False)
st = lambda it: eng.label_stats(None, K)
for name, fn in (("fused labels+stats", fused), ("labels only", lab), ("label_stats(resident)", st)):
    print(f"N={N} D={D} K={K} {name}: wall {timed(fn):.3f} ms  kernel {kernel_ms(fn)[0]:.3f} ms", flush=True)
