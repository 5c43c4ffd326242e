"""Diagnostic: the SVI driver of the linear-Gaussian experts at the shape the reference's ILR examples default to (dx = dy = 1, 50 experts —
examples/ilr/evaluate_sine.py:35,40-47) at N rows: one outer iteration = a minibatch natural-gradient step + the full-data bound.  The
bound-only pass runs on the narrow kernels since round 4 (MIMO_BOUND_PROMOTE=0: the generic request on the tile kernels).
    python tools/ilr_svi_time.py [N]"""
import os, subprocess, sys
N = sys.argv[1] if len(sys.argv) > 1 else "4e6"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, random
import numpy as np, numpy.random as npr
sys.path.insert(0, sys.argv[1])
from mimo_amd.engine import HipEngine
from mimo_amd.distributions import (TruncatedStickBreaking, CategoricalWithStickBreaking, StackedNormalWisharts, StackedGaussiansWithNormalWisharts,
                                    StackedMatrixNormalWisharts, StackedLinearGaussiansWithMatrixNormalWisharts)
from mimo_amd.mixtures import BayesianMixtureOfLinearGaussians
N, K, dx, dy = int(float(sys.argv[2])), 50, 1, 1
rng = np.random.default_rng(0)
X = rng.uniform(-6., 6., size=(N, 1)); Y = np.sin(X) + 0.1 * rng.standard_normal((N, 1))
eng = HipEngine(0)
npr.seed(1); random.seed(2)
gs = CategoricalWithStickBreaking(K, TruncatedStickBreaking(K, np.ones(K), 5. * np.ones(K)))
bp = StackedNormalWisharts(K, dx, np.zeros((K, dx)), 1e-2 * np.ones(K), np.stack(K * [np.eye(dx)]), (dx + 2.) * np.ones(K))
mp = StackedMatrixNormalWisharts(K, dx + 1, dy, np.zeros((K, dy, dx + 1)), np.stack(K * [1e-2 * np.eye(dx + 1)]), np.stack(K * [np.eye(dy)]), (dy + 2.) * np.ones(K))
m = BayesianMixtureOfLinearGaussians(K, dx, dy, gs, StackedGaussiansWithNormalWisharts(K, dx, bp, engine=eng),
                                     StackedLinearGaussiansWithMatrixNormalWisharts(K, dx + 1, dy, mp, engine=eng), engine=eng)
m.resample(X, Y, maxiter=3, progress_bar=False, label_rng='philox', seed=1)
run = lambda it: m.meanfield_stochastic_descent(X, Y, randomize=False, maxiter=it, step_size=1e-2, batch_size=4096, progress_bar=False)
run(4)
a = b = 1e9
for _ in range(3):
    t0 = time.perf_counter(); run(4); a = min(a, time.perf_counter() - t0)
for _ in range(3):
    t0 = time.perf_counter(); run(44); b = min(b, time.perf_counter() - t0)
print("RES %.4f" % ((b - a) / 40 * 1e3))
'''
for mode, name in (("0", "bound-only pass as the generic request (tile kernels)"), ("1", "bound-only pass on the narrow kernels (default)")):
    e = dict(os.environ); e["MIMO_BOUND_PROMOTE"] = mode
    r = subprocess.run([sys.executable, "-c", CHILD, root, N], capture_output=True, text=True, env=e)
    line = [l for l in r.stdout.splitlines() if l.startswith("RES")]
    print(f"ILR SVI outer iteration, N = {N}, dx = dy = 1, K = 50, {name}: {line[0].split()[1] if line else 'ERR ' + r.stderr[-300:]} ms", flush=True)
