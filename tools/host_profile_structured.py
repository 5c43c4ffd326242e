"""Diagnostic: where the host time of the structured drivers goes (cProfile of 20 iterations of the tied-covariance and the
hierarchical mean-field drivers and of 10 SVI outer iterations at N rows; the kernels run asynchronously under it).
    python tools/host_profile_structured.py [N] [tied|hier|svi|c1]"""
import cProfile, io, os, pstats, sys, time
import numpy as np
import numpy.random as npr
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd.engine import HipEngine
from mimo_amd.distributions import (Dirichlet, CategoricalWithDirichlet, StackedNormalWisharts, StackedGaussiansWithNormalWisharts,
                                    TiedNormalWisharts, TiedGaussiansWithNormalWisharts, NormalWishart,
                                    TiedGaussiansWithScaledPrecision, TiedGaussiansWithHierarchicalNormalWisharts)
from mimo_amd.mixtures import BayesianMixtureOfGaussians, BayesianMixtureOfGaussiansWithHierarchicalPrior
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000
which = sys.argv[2] if len(sys.argv) > 2 else "tied"
D, K = (2, 4) if which == "c1" else (16, 64)
eng = HipEngine(0)
rng = np.random.default_rng(3)
centres = rng.normal(0., 6., size=(min(K, 32), D))
X = np.ascontiguousarray(centres[rng.integers(len(centres), size=N)] + rng.standard_normal((N, D)))
gd = lambda: CategoricalWithDirichlet(K, Dirichlet(K, np.ones(K)))
npr.seed(1)
if which == "tied":
    prior = TiedNormalWisharts(K, D, np.zeros((K, D)), 1e-2 * np.ones(K), np.stack(K * [np.eye(D)]), (D + 2.) * np.ones(K))
    m = BayesianMixtureOfGaussians(gd(), TiedGaussiansWithNormalWisharts(K, D, prior, engine=eng), engine=eng)
    run = lambda it: m.meanfield_coordinate_descent(X, randomize=False, maxiter=it, tol=0., progress_bar=False)
elif which == "hier":
    hyper = NormalWishart(D, np.zeros(D), 1e-2, np.eye(D), D + 2.)
    hp = TiedGaussiansWithScaledPrecision(K, D, kappas=1e-2 * np.ones(K))
    m = BayesianMixtureOfGaussiansWithHierarchicalPrior(K, D, gd(), TiedGaussiansWithHierarchicalNormalWisharts(K, D, hyper, hp, engine=eng), engine=eng)
    run = lambda it: m.meanfield_coordinate_descent(X, randomize=False, maxiter=it, maxsubiter=5, tol=0., progress_bar=False)
else:
    prior = StackedNormalWisharts(K, D, np.zeros((K, D)), 1e-2 * np.ones(K), np.stack(K * [np.eye(D)]), (D + 2.) * np.ones(K))
    m = BayesianMixtureOfGaussians(gd(), StackedGaussiansWithNormalWisharts(K, D, prior, engine=eng), engine=eng)
    if which == "svi":
        run = lambda it: m.meanfield_stochastic_descent(X, randomize=False, maxiter=it, batch_size=4096, progress_bar=False)
    else:
        run = lambda it: m.meanfield_coordinate_descent(X, randomize=False, maxiter=it, tol=0., progress_bar=False)
run(4)
t0 = time.perf_counter(); run(4); a = time.perf_counter() - t0
t0 = time.perf_counter(); run(44); b = time.perf_counter() - t0
print(f"{which}: {(b - a) / 40 * 1e3:.3f} ms per iteration (N = {N}, D = {D}, K = {K})")
pr = cProfile.Profile(); pr.enable(); run(40); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
