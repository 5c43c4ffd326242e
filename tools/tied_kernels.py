"""Diagnostic: which kernels a tied-covariance VI iteration of the public driver launches, and their device time."""
import os, sys, time
import numpy as np, numpy.random as npr
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd.engine import HipEngine
from mimo_amd.distributions import Dirichlet, CategoricalWithDirichlet, TiedNormalWisharts, TiedGaussiansWithNormalWisharts
from mimo_amd.mixtures import BayesianMixtureOfGaussians
N, D, K = 4_000_000, 16, 64
rng = np.random.default_rng(0)
X = rng.standard_normal((N, D)) + 3. * rng.standard_normal((K, D))[rng.integers(0, K, N)]
eng = HipEngine(0)
npr.seed(1)
prior = TiedNormalWisharts(K, D, np.zeros((K, D)), 1e-2 * np.ones(K), np.stack(K * [np.eye(D)]), (D + 2.) * np.ones(K))
tied = BayesianMixtureOfGaussians(CategoricalWithDirichlet(K, Dirichlet(K, np.ones(K))), TiedGaussiansWithNormalWisharts(K, D, prior, engine=eng), engine=eng)
tied.meanfield_coordinate_descent(X, randomize=False, maxiter=3, tol=0., progress_bar=False)
eng.profile(True); eng.profile_read(reset=True)
t0 = time.perf_counter()
tied.meanfield_coordinate_descent(X, randomize=False, maxiter=20, tol=0., progress_bar=False)
dt = time.perf_counter() - t0
print("wall per iteration (incl. the fixed part of a call) ms:", dt / 20 * 1e3, "plan:", eng.plan(K))
for name, v in eng.profile_kernels().items():
    print(f"  {name:50s} launches {v['launches']:4d}  {v['ms'] / v['launches']:.3f} ms")
