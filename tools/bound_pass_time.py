"""Diagnostic: the bound-only pass (MIMO_F_NO_STATS, no tables) as the generic request on the tile kernels (MIMO_BOUND_PROMOTE=0) and as
the plain pass of the shape's own kernel family with the statistics left in the partial blocks (=2), per shape, in child processes.
    python tools/bound_pass_time.py [N]"""
import os, subprocess, sys
N = sys.argv[1] if len(sys.argv) > 1 else "2e6"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
from mimo_amd.engine import HipEngine
N, D, K = int(float(sys.argv[2])), int(sys.argv[3]), int(sys.argv[4])
rng = np.random.default_rng(0)
eng = HipEngine(0)
eng.upload(rng.standard_normal((N, D)) * 2.)
A = rng.standard_normal((K, D, D)); W = A @ A.transpose(0, 2, 1) / D + np.eye(D)
b = rng.standard_normal((K, D)); c = rng.standard_normal(K)
def t(fn, n=12):
    fn(); fn(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3
full = t(lambda: eng.estep(c, b, W)); bound = t(lambda: eng.estep(c, b, W, stats=False))
s1 = eng.estep(c, b, W)[1][0]; s2 = eng.estep(c, b, W, stats=False)[1][0]
print("RES %s %.4f %.4f %.3e" % (eng.plan(K)["kind"], full, bound, abs(s1 - s2) / abs(s1)))
'''
shapes = [(1, 100), (2, 50), (2, 128), (2, 160), (3, 64), (4, 64), (4, 128), (8, 4), (8, 16), (12, 8), (16, 4), (16, 16), (24, 4), (32, 4),
          (20, 16), (24, 16), (32, 16), (24, 32), (32, 32), (28, 48), (16, 40), (16, 48), (12, 40), (16, 80), (20, 96), (12, 112), (8, 64), (16, 64)]
print("# N = %s; ms per pass: full plain pass | bound-only, generic request | bound-only as the plain pass" % N, flush=True)
for D, K in shapes:
    out = []
    for mode in ("0", "2"):
        e = dict(os.environ); e["MIMO_BOUND_PROMOTE"] = mode
        r = subprocess.run([sys.executable, "-c", CHILD, root, N, str(D), str(K)], capture_output=True, text=True, env=e)
        line = [l for l in r.stdout.splitlines() if l.startswith("RES")]
        out.append(line[0].split() if line else ["RES", "ERR", "nan", "nan", r.stderr[-150:]])
    print(f"Dz={D:2d} K={K:3d} [{out[0][1]:10s}] full {float(out[0][2]):.3f} | generic {float(out[0][3]):.3f} | promoted {float(out[1][3]):.3f}   (agreement {out[1][4]})", flush=True)
