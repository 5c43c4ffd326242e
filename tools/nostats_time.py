"""Diagnostic: the bound-only data pass (MIMO_F_NO_STATS: log-densities + log-sum-exp, what the SVI drivers run over the
full data every outer iteration) against the full mean-field pass, same rows and parameters.
    python tools/nostats_time.py [N] [D] [K]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd.engine import HipEngine
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 16
K = int(sys.argv[3]) if len(sys.argv) > 3 else 64
rng = np.random.default_rng(0)
eng = HipEngine(0)
X = rng.standard_normal((N, D)) * 2.
eng.upload(X)
A = rng.standard_normal((K, D, D))
W = A @ A.transpose(0, 2, 1) / D + np.eye(D)
b = rng.standard_normal((K, D))
c = rng.standard_normal(K)
def t(fn, n=20):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e3
full = t(lambda: eng.estep(c, b, W))
bound = t(lambda: eng.estep(c, b, W, stats=False))
def pipelined():
    eng.estep_async(c, b, W, stats=False); return eng.estep_wait()
boundp = t(pipelined)
flops_l = 2. * N * K * ((D + 1) * (D + 2) / 2)
print(f"N={N} D={D} K={K}: full pass {full:.3f} ms ({2 * flops_l / full / 1e9 / 78.6:.3f} of 78.6 TF), bound-only pass {bound:.3f} ms "
      f"({flops_l / bound / 1e9 / 78.6:.3f}), async form {boundp:.3f} ms; plan {eng.plan(K)}")
S, sc = eng.estep(c, b, W)
_, sc2 = eng.estep(c, b, W, stats=False)
print("bound terms agree:", sc[0], sc2[0], abs(sc[0] - sc2[0]) / abs(sc[0]))
