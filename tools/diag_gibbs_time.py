import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.getcwd())
import numpy as np, numpy.random as npr
from mimo_amd.engine import HipEngine
from mimo_amd.distributions import Dirichlet, CategoricalWithDirichlet, StackedNormalGammas, StackedGaussiansWithNormalGammas
from mimo_amd.mixtures import BayesianMixtureOfGaussians
N, D, K = 4_000_000, 16, 64
eng = HipEngine(0)
rng = np.random.default_rng(3)
centres = rng.normal(0., 6., size=(32, D))
X = np.ascontiguousarray(centres[rng.integers(32, size=N)] + rng.standard_normal((N, D)))
npr.seed(1)
prior = StackedNormalGammas(K, D, np.zeros((K, D)), 1e-2 * np.ones((K, D)), (D + 1.) / 2. * np.ones((K, D)), 0.5 * np.ones((K, D)))
diag = BayesianMixtureOfGaussians(CategoricalWithDirichlet(K, Dirichlet(K, np.ones(K))), StackedGaussiansWithNormalGammas(K, D, prior, engine=eng), engine=eng)
for vi_first in (False, True):
    if vi_first:
        diag.meanfield_coordinate_descent(X, randomize=False, maxiter=30, tol=0., progress_bar=False)
    run = lambda it: diag.resample(X, maxiter=it, progress_bar=False, label_rng='philox', seed=1)
    run(4)
    t0 = time.perf_counter(); run(4); a = time.perf_counter() - t0
    t0 = time.perf_counter(); run(44); b = time.perf_counter() - t0
    print("vi_first", vi_first, "ms/iter %.3f" % ((b - a) / 40 * 1e3), "plan", eng.plan(K), flush=True)
    pr = cProfile.Profile(); pr.enable(); run(40); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(10); print(s.getvalue()[:2500])
