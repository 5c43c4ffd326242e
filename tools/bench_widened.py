"""Timings of the widened paths through the host API, at large N on one GPU:

    python tools/bench_widened.py [N]        (default N = 4e6 rows, D = 16, K = 64; ILR dx = 8, dy = 4)

For every driver the cost of ONE iteration is the difference between a long and a short run divided by the
extra iterations (upload and initialisation cancel).  Prints one JSON line per path:
rows x components evaluated per second, and the same kernel-level FP64 figure bench.py reports
(algorithmic flops of SURVEY.md section 8(d) / 78.6 TFLOP/s) where the path is the fused E-step + statistics pass.
"""
import json
import os
import sys
import time

import numpy as np
import numpy.random as npr

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mimo_amd.engine import HipEngine
from mimo_amd.distributions import (Dirichlet, TruncatedStickBreaking, CategoricalWithDirichlet, CategoricalWithStickBreaking,
                                    StackedNormalWisharts, StackedGaussiansWithNormalWisharts, TiedNormalWisharts,
                                    TiedGaussiansWithNormalWisharts, StackedNormalGammas, StackedGaussiansWithNormalGammas,
                                    NormalWishart, TiedGaussiansWithScaledPrecision, TiedGaussiansWithHierarchicalNormalWisharts,
                                    StackedMatrixNormalWisharts, StackedLinearGaussiansWithMatrixNormalWisharts)
from mimo_amd.mixtures import (BayesianMixtureOfGaussians, BayesianMixtureOfGaussiansWithHierarchicalPrior,
                               BayesianMixtureOfLinearGaussians)

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000
D, K = 16, 64
PEAK = 78.6e12
eng = HipEngine(0)
rng = np.random.default_rng(3)
centres = rng.normal(0., 6., size=(32, D))
X = np.ascontiguousarray(centres[rng.integers(32, size=N)] + rng.standard_normal((N, D)))


def per_iter(run, short=4, long=24):
    run(short)                                   # warm-up (upload, allocations)
    a = b = float("inf")                         # best of three each: the fixed part of a run (label initialisation over N rows,
    for _ in range(3):                           # fingerprints) jitters by more than a few iterations cost
        t0 = time.perf_counter(); run(short); a = min(a, time.perf_counter() - t0)
    for _ in range(3):
        t0 = time.perf_counter(); run(long); b = min(b, time.perf_counter() - t0)
    return (b - a) / (long - short)


def report(path, t, evals, flops=None, note=""):
    out = {"path": path, "rows": N, "ms_per_iteration": 1e3 * t, "evals_per_s": evals / t}
    if flops is not None:
        out["fp64_frac_of_78.6TF"] = flops / t / PEAK
    if note:
        out["note"] = note
    print(json.dumps(out), flush=True)


FE, FS = D * (D + 1) + 3 * D + 8, (D + 1) * (D + 2) + 1
gd = lambda: CategoricalWithDirichlet(K, Dirichlet(K, np.ones(K)))

npr.seed(1)
prior = StackedNormalWisharts(K, D, np.zeros((K, D)), 1e-2 * np.ones(K), np.stack(K * [np.eye(D)]), (D + 2.) * np.ones(K))
full = BayesianMixtureOfGaussians(gd(), StackedGaussiansWithNormalWisharts(K, D, prior, engine=eng), engine=eng)
t = per_iter(lambda it: full.meanfield_coordinate_descent(X, randomize=False, maxiter=it, tol=0., progress_bar=False))
report("full-covariance GMM, mean-field VI (reference-shaped driver incl. likelihood refresh)", t, N * K, N * K * (FE + FS))

t = per_iter(lambda it: full.meanfield_coordinate_descent(X, randomize=False, maxiter=it, tol=0., progress_bar=False,
                                                          sample_likelihood=False))
report("full-covariance GMM, mean-field VI, sample_likelihood=False (no per-iteration posterior.rvs())", t, N * K,
       N * K * (FE + FS))

prior = TiedNormalWisharts(K, D, np.zeros((K, D)), 1e-2 * np.ones(K), np.stack(K * [np.eye(D)]), (D + 2.) * np.ones(K))
tied = BayesianMixtureOfGaussians(gd(), TiedGaussiansWithNormalWisharts(K, D, prior, engine=eng), engine=eng)
t = per_iter(lambda it: tied.meanfield_coordinate_descent(X, randomize=False, maxiter=it, tol=0., progress_bar=False))
FEl, FSl = 2 * (D + 1) + 8, 2 * (D + 1) + 1              # 'linear' structure: both products over the D+1 features z_a, 1
report("tied-covariance GMM, mean-field VI (MIMO_STRUCT_LINEAR: D+1 feature kernels)", t, N * K, N * K * (FEl + FSl),
       "flops of the linear form (the shared quadratic term never reaches the kernels): 2(D+1)+8 and 2(D+1)+1 per evaluation")

prior = StackedNormalGammas(K, D, np.zeros((K, D)), 1e-2 * np.ones((K, D)), (D + 1.) / 2. * np.ones((K, D)), 0.5 * np.ones((K, D)))
diag = BayesianMixtureOfGaussians(gd(), StackedGaussiansWithNormalGammas(K, D, prior, engine=eng), engine=eng)
t = per_iter(lambda it: diag.meanfield_coordinate_descent(X, randomize=False, maxiter=it, tol=0., progress_bar=False))
FEd, FSd = 2 * (2 * D + 1) + 8, 2 * (2 * D + 1) + 1      # both products over the 2D+1 features z_a^2, z_a, 1
report("diagonal-precision GMM, mean-field VI (MIMO_STRUCT_DIAG: 2D+1 feature kernels; reference-shaped driver)", t,
       N * K, N * K * (FEd + FSd), "flops of the diagonal form: 2(2D+1)+8 and 2(2D+1)+1 per evaluation; the iteration "
       "includes ~1 ms of host posterior.rvs()")
# the sweeps themselves (gibbs_iteration, what `resample` loops over): the driver's label initialisation draws N labels on the
# host once per call (0.1 s at 4e6 rows) and its jitter is ten sweeps' worth — the per_iter difference swung 0.4 - 1.1 ms on it
diag.resample(X, maxiter=2, progress_bar=False, label_rng='philox', seed=1, init_labels='posterior')
_eng = diag._bind(X)
_state = [_eng.label_stats(diag.labels_, K)]


def _sweeps(n):
    for it in range(n):
        _state[0] = diag.gibbs_iteration(_eng, _state[0], it + 1, 'philox', 1)[1]


_sweeps(5)
t0 = time.perf_counter(); _sweeps(60); t = (time.perf_counter() - t0) / 60
report("diagonal-precision GMM, Gibbs sweep (Philox labels)", t, N * K, N * K * FEd + N * FSd)

# full-covariance Gibbs sweeps with the posterior draws in the reference's numpy.random order (seeded runs reproduce the reference's
# chain) against the batched draws from a Generator: since round 4 the former are one native call per sweep (mimo_host_legacy_draws)
full.resample(X, maxiter=2, progress_bar=False, label_rng='philox', seed=1, init_labels='posterior')
for name, prng in (("numpy.random in the reference's per-component order", None),
                   ("batched draws from a numpy Generator", np.random.Generator(np.random.Philox(7)))):
    _feng = full._bind(X)
    st = [_feng.label_stats(full.labels_, K)]

    def _fs(n):
        for it in range(n):
            st[0] = full.gibbs_iteration(_feng, st[0], it + 1, 'philox', 1, prng)[1]

    _fs(5)
    t0 = time.perf_counter(); _fs(40); t = (time.perf_counter() - t0) / 40
    report("full-covariance GMM, Gibbs sweep (Philox labels), posterior draws: " + name, t, N * K, N * K * FE + N * FS)

hyper = NormalWishart(D, np.zeros(D), 1e-2, np.eye(D), D + 2.)
hp = TiedGaussiansWithScaledPrecision(K, D, kappas=1e-2 * np.ones(K))
hier = BayesianMixtureOfGaussiansWithHierarchicalPrior(
    K, D, gd(), TiedGaussiansWithHierarchicalNormalWisharts(K, D, hyper, hp, engine=eng), engine=eng)
t = per_iter(lambda it: hier.meanfield_coordinate_descent(X, randomize=False, maxiter=it, maxsubiter=5, tol=0.,
                                                          progress_bar=False))
report("hierarchical (Normal-Wishart hyper-prior) GMM, mean-field VI, 5 sub-iterations (linear structure)", t, N * K,
       N * K * (FEl + FSl), "flops of the linear form")
w = np.linspace(0.5, 1., N)
t = per_iter(lambda it: hier.meanfield_coordinate_descent(X, randomize=False, weights=w, maxiter=it, maxsubiter=5, tol=0.,
                                                          progress_bar=False))
report("hierarchical GMM, mean-field VI with per-row weights (mimo_estep_weighted, linear structure)", t, N * K,
       N * K * (FEl + FSl), "flops of the linear form")

t = per_iter(lambda it: full.meanfield_stochastic_descent(X, randomize=False, maxiter=it, batch_size=4096, progress_bar=False))
report("full-covariance GMM, SVI outer iteration (4096-row natural-gradient step + full-data bound)", t, N * K,
       N * K * FE, "the full-data pass skips the statistics")

dx, dy, Ki = 8, 4, 64
Xi = np.ascontiguousarray(X[:, :dx])
A = rng.normal(size=(32, dy, dx)) / np.sqrt(dx)
lab = rng.integers(32, size=N)
Y = np.einsum('ndl,nl->nd', A[lab], Xi) + 0.3 * rng.standard_normal((N, dy))
npr.seed(2)
gs = CategoricalWithStickBreaking(Ki, TruncatedStickBreaking(Ki, np.ones(Ki), 5. * np.ones(Ki)))
bprior = StackedNormalWisharts(Ki, dx, np.zeros((Ki, dx)), 1e-2 * np.ones(Ki), np.stack(Ki * [1e2 * np.eye(dx)]),
                               (dx + 2.) * np.ones(Ki))
mprior = StackedMatrixNormalWisharts(Ki, dx + 1, dy, np.zeros((Ki, dy, dx + 1)), np.stack(Ki * [1e-2 * np.eye(dx + 1)]),
                                     np.stack(Ki * [np.eye(dy)]), (dy + 2.) * np.ones(Ki))
ilr = BayesianMixtureOfLinearGaussians(Ki, dx, dy, gs, StackedGaussiansWithNormalWisharts(Ki, dx, bprior, engine=eng),
                                       StackedLinearGaussiansWithMatrixNormalWisharts(Ki, dx + 1, dy, mprior, engine=eng),
                                       engine=eng)
ilr.meanfield_coordinate_descent(Xi, Y, randomize=False, maxiter=3, tol=0., progress_bar=False)
ilr.meanfield_prediction(Xi[:1000])
ts = []
for _ in range(3):
    t0 = time.perf_counter(); ilr.meanfield_prediction(Xi); ts.append(time.perf_counter() - t0)
report("ILR posterior-predictive mixture moments (meanfield_prediction, average; host tables in and out)",
       min(ts), N * Ki, None, "includes the PCIe upload of x (N x 8) and the download of mu / covar (N x 20)")
