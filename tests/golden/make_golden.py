"""Generate golden vectors by importing the REFERENCE (hanyas/mimo) in the build container.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden.py

Writes tests/golden/*.npz (inputs and the reference's outputs only — no reference code).  The
reference lives at /root/reference, is read-only, and never travels to the GPU box; the committed
.npz files do.  Seeds are fixed, so the files are reproducible bit-for-bit on this image
(numpy 2.2.6 / scipy 1.15.3 / OpenBLAS 0.3.29).
"""
import copy
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

import numpy as np
import numpy.random as npr

import mimo.utils.stats as ref_stats
from mimo.distributions import (Dirichlet, TruncatedStickBreaking, CategoricalWithDirichlet,
                                CategoricalWithStickBreaking, StackedNormalWisharts,
                                StackedGaussiansWithNormalWisharts, StackedMatrixNormalWisharts,
                                StackedLinearGaussiansWithMatrixNormalWisharts)
from mimo.mixtures import BayesianMixtureOfGaussians, BayesianMixtureOfLinearGaussians
from mimo.utils.data import one_hot

from oracle.mimo_oracle import philox_uniforms   # only the counter-based uniforms (not in the reference)

PHILOX_SEED, PHILOX_SWEEP = 1337, 3


class FixedUniforms:
    """Shim for the single npr.random((1,N)) call of sample_discrete_from_log (stats.py:14)."""

    def __init__(self, u):
        self.u = u

    def __enter__(self):
        self._orig = ref_stats.npr.random
        ref_stats.npr.random = lambda size=None: np.reshape(self.u, size)
        return self

    def __exit__(self, *a):
        ref_stats.npr.random = self._orig


def make_data(N, D, n_clusters=4):
    A = npr.randn(D, D) / np.sqrt(D)
    shifts = 4.0 * npr.randn(n_clusters, D)
    X = npr.randn(N, D) @ A + shifts[npr.randint(n_clusters, size=N)]
    return np.ascontiguousarray(X)


def make_gating(K, kind):
    if kind == 'dirichlet':
        prior = Dirichlet(dim=K, alphas=np.ones((K,)))
        return CategoricalWithDirichlet(dim=K, prior=prior)
    prior = TruncatedStickBreaking(dim=K, gammas=np.ones((K,)), deltas=5. * np.ones((K,)))
    return CategoricalWithStickBreaking(dim=K, prior=prior)


def gating_params(g, kind):
    if kind == 'dirichlet':
        return dict(alphas=np.array(g.alphas, dtype=float))
    return dict(gammas=np.array(g.gammas, dtype=float), deltas=np.array(g.deltas, dtype=float))


def put(out, prefix, d):
    for k, v in d.items():
        out[f"{prefix}_{k}"] = np.asarray(v)


def nw_params(nw):
    return dict(mus=nw.mus, kappas=nw.kappas, psis=nw.psis, nus=nw.nus)


def mnw_params(p):
    return dict(Ms=p.Ms, Ks=p.Ks, psis=p.psis, nus=p.nus)


def gmm_case(name, N, D, K, kind, seed, vi_iters=20):
    npr.seed(seed)
    X = make_data(N, D)
    gating = make_gating(K, kind)
    prior = StackedNormalWisharts(size=K, dim=D, mus=np.zeros((K, D)), kappas=1e-2 * np.ones((K,)),
                                  psis=np.stack(K * [np.eye(D)]), nus=(D + 1.) * np.ones((K,)) + 1e-8)
    comps = StackedGaussiansWithNormalWisharts(size=K, dim=D, prior=prior)
    model = BayesianMixtureOfGaussians(gating=gating, components=comps)

    out = dict(X=X, gating_kind=np.array(kind), K=np.array(K), D=np.array(D))
    put(out, "prior", nw_params(model.components.prior))
    put(out, "gprior", gating_params(model.gating.prior, kind))

    # posterior from one M-step on seeded random responsibilities (gmm.py:265-267, 289-291)
    resp0 = npr.rand(K, N)
    resp0 /= np.sum(resp0, axis=0)
    out["resp0"] = resp0
    st0 = model.components.likelihood.weighted_statistics(X.copy(), resp0)          # A10
    put(out, "stats0", dict(xk=st0[0], nk=st0[1], xxTk=st0[2]))
    out["counts0"] = model.gating.likelihood.weighted_statistics(None, resp0)       # A12
    model.meanfield_update_parameters(X.copy(), resp0)                              # A13 (+ rvs)
    put(out, "post", nw_params(model.components.posterior))
    put(out, "gpost", gating_params(model.gating.posterior, kind))
    put(out, "lik", dict(mus=model.components.likelihood.mus, lmbdas=model.components.likelihood.lmbdas,
                         probs=model.gating.likelihood.probs))

    # Gibbs / EM form (A1, A2)
    out["A1_loglik"] = model.components.likelihood.log_likelihood(X.copy())
    out["A2_lcl"] = model.likelihood.log_complete_likelihood(X.copy())
    out["A2_resp"] = model.likelihood.responsibilities(X.copy())
    out["A2_ll"] = model.likelihood.log_likelihood(X.copy())

    # VI form (A3, A4) and the expectations that feed it
    es = model.components.posterior.expected_statistics()
    put(out, "estats", dict(a=es[0], b=es[1], c=es[2], d=es[3]))
    out["A3_eloglik"] = model.components.expected_log_likelihood(X.copy())
    out["A4_elcl"] = model.expected_log_complete_likelihood(X.copy())
    eresp = model.expected_responsibilities(X.copy())
    out["A4_eresp"] = eresp
    out["A4_ell"] = model.expected_log_likelihood(X.copy())

    # statistics of the VI responsibilities (A10, A12)
    st = model.components.likelihood.weighted_statistics(X.copy(), eresp)
    put(out, "stats", dict(xk=st[0], nk=st[1], xxTk=st[2]))
    out["counts"] = model.gating.likelihood.weighted_statistics(None, eresp)

    # ELBO terms at the current posterior (A14)
    out["vlb_obs"] = model.variational_lowerbound_obs(X.copy(), eresp)
    out["vlb_labels"] = model.variational_lowerbound_labels(eresp)
    out["vlb_gating"] = model.gating.variational_lowerbound()
    out["vlb_comps"] = model.components.variational_lowerbound()
    out["vlb_total"] = model.variational_lowerbound(X.copy(), eresp)

    # label draw (A8) with captured uniforms: MT19937 uniforms and the engine's Philox uniforms
    u_mt = npr.random(size=(1, N))
    with FixedUniforms(u_mt):
        _, labels_mt = model.resample_labels(X.copy())
    u_ph = philox_uniforms(PHILOX_SEED, np.arange(N), PHILOX_SWEEP)
    with FixedUniforms(u_ph):
        _, labels_ph = model.resample_labels(X.copy())
    out["u_mt"], out["labels_mt"], out["labels_philox"] = u_mt.ravel(), labels_mt, labels_ph
    oh = one_hot(labels_mt, K)                                                       # A9
    stl = model.components.likelihood.weighted_statistics(X.copy(), oh)
    put(out, "lstats", dict(xk=stl[0], nk=stl[1], xxTk=stl[2]))
    out["lcounts"] = model.gating.likelihood.statistics(labels_mt)

    # conjugate update from the VI responsibilities (A13) on a copy
    m2 = copy.deepcopy(model)
    m2.meanfield_update_parameters(X.copy(), eresp)
    put(out, "post2", nw_params(m2.components.posterior))
    put(out, "gpost2", gating_params(m2.gating.posterior, kind))

    # (7) ELBO trace of meanfield_coordinate_descent(randomize=False) from this posterior
    m3 = copy.deepcopy(model)
    vlb = m3.meanfield_coordinate_descent(X.copy(), randomize=False, maxiter=vi_iters, tol=0., progress_bar=False)
    out["vi_vlb"] = np.array(vlb)
    put(out, "vi_post", nw_params(m3.components.posterior))
    put(out, "vi_gpost", gating_params(m3.gating.posterior, kind))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok", {k: v.shape for k, v in out.items() if hasattr(v, 'shape') and v.size > 4096})


def gibbs_trace_case(name, N, D, K, kind, seed, sweeps=5):
    """(8) seeded Gibbs trace pinning the host-RNG call order of gmm.py:207-225."""
    npr.seed(seed)
    X = make_data(N, D)
    gating = make_gating(K, kind)
    prior = StackedNormalWisharts(size=K, dim=D, mus=np.zeros((K, D)), kappas=1e-2 * np.ones((K,)),
                                  psis=np.stack(K * [np.eye(D)]), nus=(D + 1.) * np.ones((K,)) + 1e-8)
    comps = StackedGaussiansWithNormalWisharts(size=K, dim=D, prior=prior)
    model = BayesianMixtureOfGaussians(gating=gating, components=comps)
    out = dict(X=X, gating_kind=np.array(kind), K=np.array(K), D=np.array(D), seed2=np.array(seed + 1))
    put(out, "prior", nw_params(model.components.prior))
    put(out, "gprior", gating_params(model.gating.prior, kind))
    put(out, "lik0", dict(mus=model.components.likelihood.mus, lmbdas=model.components.likelihood.lmbdas,
                          probs=model.gating.likelihood.probs))
    mA = copy.deepcopy(model)
    # explicit loop (same body as gmm.py:220-223), recording every sweep
    npr.seed(seed + 1)
    labels = mA.gating.likelihood.rvs(len(X))                 # init_labels='prior' (gmm.py:213)
    out["labels_init"] = labels
    for s in range(sweeps):
        mA.resample_components(X.copy(), labels)
        mA.resample_gating(labels)
        _, labels = mA.resample_labels(X.copy())
        out[f"s{s}_labels"] = labels
        out[f"s{s}_mus"] = mA.components.likelihood.mus
        out[f"s{s}_lmbdas"] = mA.components.likelihood.lmbdas
        out[f"s{s}_probs"] = np.array(mA.gating.likelihood.probs)
        put(out, f"s{s}_post", nw_params(mA.components.posterior))
    # the driver itself, same seed: final state must coincide with the explicit loop
    mB = copy.deepcopy(model)
    npr.seed(seed + 1)
    mB.resample(X.copy(), init_labels='prior', maxiter=sweeps, progress_bar=False)
    out["driver_mus"] = mB.components.likelihood.mus
    out["driver_probs"] = np.array(mB.gating.likelihood.probs)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok")


def ilr_case(name, N, dx, dy, K, kind, seed, vi_iters=20):
    npr.seed(seed)
    X = make_data(N, dx)
    Atrue = npr.randn(4, dy, dx)
    Y = np.einsum('ndl,nl->nd', Atrue[npr.randint(4, size=N)], X) + 0.3 * npr.randn(N, dy)
    Y = np.ascontiguousarray(Y)
    dc = dx + 1
    gating = make_gating(K, kind)
    bprior = StackedNormalWisharts(size=K, dim=dx, mus=np.zeros((K, dx)), kappas=1e-2 * np.ones((K,)),
                                   psis=np.stack(K * [1e2 * np.eye(dx)]), nus=(dx + 1.) * np.ones((K,)) + 1e-16)
    basis = StackedGaussiansWithNormalWisharts(size=K, dim=dx, prior=bprior)
    mprior = StackedMatrixNormalWisharts(K, dc, dy, Ms=np.zeros((K, dy, dc)), Ks=np.stack(K * [1e-2 * np.eye(dc)]),
                                         psis=np.stack(K * [np.eye(dy)]), nus=(dy + 1.) * np.ones((K,)) + 1e-16)
    models = StackedLinearGaussiansWithMatrixNormalWisharts(K, dc, dy, mprior, affine=True)
    ilr = BayesianMixtureOfLinearGaussians(size=K, input_dim=dx, output_dim=dy,
                                           gating=gating, basis=basis, models=models)
    out = dict(X=X, Y=Y, gating_kind=np.array(kind), K=np.array(K))
    put(out, "bprior", nw_params(ilr.basis.prior))
    put(out, "mprior", mnw_params(ilr.models.prior))
    put(out, "gprior", gating_params(ilr.gating.prior, kind))

    resp0 = npr.rand(K, N)
    resp0 /= np.sum(resp0, axis=0)
    out["resp0"] = resp0
    ms0 = ilr.models.likelihood.weighted_statistics(X.copy(), Y.copy(), resp0)      # A11
    put(out, "mstats0", dict(yxTk=ms0[0], xxTk=ms0[1], yyTk=ms0[2], nk=ms0[3]))
    ilr.meanfield_update_parameters(X.copy(), Y.copy(), resp0)
    put(out, "bpost", nw_params(ilr.basis.posterior))
    put(out, "mpost", mnw_params(ilr.models.posterior))
    put(out, "gpost", gating_params(ilr.gating.posterior, kind))
    put(out, "lik", dict(mus=ilr.basis.likelihood.mus, lmbdas=ilr.basis.likelihood.lmbdas,
                         As=ilr.models.likelihood.As, lmbdas_y=ilr.models.likelihood.lmbdas,
                         probs=ilr.gating.likelihood.probs))

    out["A5_loglik"] = ilr.models.likelihood.log_likelihood(X.copy(), Y.copy())
    out["A7_lcl"] = ilr.likelihood.log_complete_likelihood(X.copy(), Y.copy())
    out["A7_resp"] = ilr.likelihood.responsibilities(X.copy(), Y.copy())
    es = ilr.models.posterior.expected_statistics()
    put(out, "mestats", dict(a=es[0], b=es[1], c=es[2], d=es[3]))
    out["A6_eloglik"] = ilr.models.expected_log_likelihood(X.copy(), Y.copy())
    out["A3_basis_eloglik"] = ilr.basis.expected_log_likelihood(X.copy())
    out["A7_elcl"] = ilr.expected_log_complete_likelihood(X.copy(), Y.copy())
    eresp = ilr.expected_responsibilities(X.copy(), Y.copy())
    out["A7_eresp"] = eresp

    bs = ilr.basis.likelihood.weighted_statistics(X.copy(), eresp)
    put(out, "bstats", dict(xk=bs[0], nk=bs[1], xxTk=bs[2]))
    ms = ilr.models.likelihood.weighted_statistics(X.copy(), Y.copy(), eresp)
    put(out, "mstats", dict(yxTk=ms[0], xxTk=ms[1], yyTk=ms[2], nk=ms[3]))
    out["counts"] = ilr.gating.likelihood.weighted_statistics(None, eresp)

    out["vlb_data"] = ilr.variational_lowerbound_data(X.copy(), Y.copy(), eresp)
    out["vlb_labels"] = ilr.variational_lowerbound_labels(eresp)
    out["vlb_gating"] = ilr.gating.variational_lowerbound()
    out["vlb_basis"] = ilr.basis.variational_lowerbound()
    out["vlb_models"] = ilr.models.variational_lowerbound()
    out["vlb_total"] = ilr.variational_lowerbound(X.copy(), Y.copy(), eresp)

    u_mt = npr.random(size=(1, N))
    with FixedUniforms(u_mt):
        _, labels_mt = ilr.resample_labels(X.copy(), Y.copy())
    u_ph = philox_uniforms(PHILOX_SEED, np.arange(N), PHILOX_SWEEP)
    with FixedUniforms(u_ph):
        _, labels_ph = ilr.resample_labels(X.copy(), Y.copy())
    out["u_mt"], out["labels_mt"], out["labels_philox"] = u_mt.ravel(), labels_mt, labels_ph

    m2 = copy.deepcopy(ilr)
    m2.meanfield_update_parameters(X.copy(), Y.copy(), eresp)
    put(out, "bpost2", nw_params(m2.basis.posterior))
    put(out, "mpost2", mnw_params(m2.models.posterior))
    put(out, "gpost2", gating_params(m2.gating.posterior, kind))

    m3 = copy.deepcopy(ilr)
    vlb = m3.meanfield_coordinate_descent(X.copy(), Y.copy(), randomize=False, maxiter=vi_iters, tol=0.,
                                          progress_bar=False)
    out["vi_vlb"] = np.array(vlb)
    put(out, "vi_bpost", nw_params(m3.basis.posterior))
    put(out, "vi_mpost", mnw_params(m3.models.posterior))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok")


def driver_traces_case(name, N, D, K, kind, seed, iters=8):
    """EM (gmm.py:77-103), MAP (gmm.py:176-204) and SVI (gmm.py:300-326) traces with fixed seeds
    (numpy.random for responsibilities / parameter draws, random for the minibatch indices)."""
    import random
    from mimo.distributions import Categorical, StackedGaussiansWithPrecision
    from mimo.mixtures import MixtureOfGaussians
    npr.seed(seed)
    X = make_data(N, D)
    out = dict(X=X, gating_kind=np.array(kind), K=np.array(K), D=np.array(D), seed=np.array(seed), iters=np.array(iters))
    # EM from seeded random responsibilities
    lik = MixtureOfGaussians(gating=Categorical(dim=K), components=StackedGaussiansWithPrecision(size=K, dim=D))
    npr.seed(seed + 10)
    out["em_loglik"] = np.array(lik.max_likelihood(X.copy(), randomize=True, maxiter=iters, progress_bar=False))
    out["em_mus"], out["em_lmbdas"], out["em_probs"] = lik.components.mus, lik.components.lmbdas, np.array(lik.gating.probs)

    def fresh():
        gating = make_gating(K, kind)
        prior = StackedNormalWisharts(size=K, dim=D, mus=np.zeros((K, D)), kappas=1e-2 * np.ones((K,)),
                                      psis=np.stack(K * [np.eye(D)]), nus=(D + 3.) * np.ones((K,)))
        comps = StackedGaussiansWithNormalWisharts(size=K, dim=D, prior=prior)
        return BayesianMixtureOfGaussians(gating=gating, components=comps)
    npr.seed(seed + 20)
    m = fresh()
    put(out, "prior", nw_params(m.components.prior))
    put(out, "gprior", gating_params(m.gating.prior, kind))
    if kind == 'dirichlet':
        m.gating.prior.alphas = 2. * np.ones(K)          # mode() needs alphas > 1
        m.gating.posterior.alphas = 2. * np.ones(K)
        out["gprior_alphas"] = m.gating.prior.alphas
        npr.seed(seed + 21)
        out["map_logprob"] = np.array(m.max_aposteriori(X.copy(), randomize=True, maxiter=iters, progress_bar=False))
        out["map_mus"] = m.components.likelihood.mus
    # SVI
    npr.seed(seed + 30)
    m = fresh()
    if kind == 'dirichlet':
        m.gating.prior.alphas = 2. * np.ones(K)
        m.gating.posterior.alphas = 2. * np.ones(K)
    npr.seed(seed + 31); random.seed(seed + 32)
    out["svi_vlb"] = np.array(m.meanfield_stochastic_descent(X.copy(), randomize=True, maxiter=iters, step_size=5e-2,
                                                             batch_size=64, progress_bar=False))
    put(out, "svi_post", nw_params(m.components.posterior))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok")


def ilr_svi_case(name, N, dx, dy, K, seed, iters=6):
    """ILR SVI trace (ilr.py:245-277) — what examples/ilr/evaluate_*.py run by default."""
    import random
    npr.seed(seed)
    X = make_data(N, dx)
    Atrue = npr.randn(4, dy, dx)
    Y = np.ascontiguousarray(np.einsum('ndl,nl->nd', Atrue[npr.randint(4, size=N)], X) + 0.3 * npr.randn(N, dy))
    dc = dx + 1
    gating = make_gating(K, 'stick')
    bprior = StackedNormalWisharts(size=K, dim=dx, mus=np.zeros((K, dx)), kappas=1e-2 * np.ones((K,)),
                                   psis=np.stack(K * [1e2 * np.eye(dx)]), nus=(dx + 1.) * np.ones((K,)) + 1e-16)
    basis = StackedGaussiansWithNormalWisharts(size=K, dim=dx, prior=bprior)
    mprior = StackedMatrixNormalWisharts(K, dc, dy, Ms=np.zeros((K, dy, dc)), Ks=np.stack(K * [1e-2 * np.eye(dc)]),
                                         psis=np.stack(K * [np.eye(dy)]), nus=(dy + 1.) * np.ones((K,)) + 1e-16)
    models = StackedLinearGaussiansWithMatrixNormalWisharts(K, dc, dy, mprior, affine=True)
    ilr = BayesianMixtureOfLinearGaussians(size=K, input_dim=dx, output_dim=dy, gating=gating, basis=basis, models=models)
    out = dict(X=X, Y=Y, gating_kind=np.array('stick'), K=np.array(K), seed=np.array(seed), iters=np.array(iters))
    put(out, "bprior", nw_params(ilr.basis.prior)); put(out, "mprior", mnw_params(ilr.models.prior))
    put(out, "gprior", gating_params(ilr.gating.prior, 'stick'))
    ilr.init_transform(X, Y)
    npr.seed(seed + 1); random.seed(seed + 2)
    ilr.resample(X.copy(), Y.copy(), init_labels='random', maxiter=3, progress_bar=False)
    out["gibbs_probs"] = np.array(ilr.gating.likelihood.probs)
    out["gibbs_As"] = ilr.models.likelihood.As
    vlb = ilr.meanfield_stochastic_descent(X.copy(), Y.copy(), randomize=False, maxiter=iters, step_size=5e-1,
                                           batch_size=64, progress_bar=False)
    out["svi_vlb"] = np.array(vlb)
    put(out, "svi_mpost", mnw_params(ilr.models.posterior))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok")


def tied_gmm_case(name, N, D, K, seed, iters=6):
    """Tied-covariance GMM (composite.py:259-283, bayesian.py:326-340, gaussian.py:545-572): seeded Gibbs
    sweeps, a VI trace and an EM trace."""
    from mimo.distributions import (TiedNormalWisharts, TiedGaussiansWithNormalWisharts, TiedGaussiansWithPrecision,
                                    Categorical)
    from mimo.mixtures import MixtureOfGaussians
    npr.seed(seed)
    X = make_data(N, D)
    out = dict(X=X, K=np.array(K), D=np.array(D), seed=np.array(seed), iters=np.array(iters), gating_kind=np.array('dirichlet'))
    gating = make_gating(K, 'dirichlet')
    prior = TiedNormalWisharts(size=K, dim=D, mus=np.zeros((K, D)), kappas=1e-2 * np.ones((K,)),
                               psis=np.stack(K * [np.eye(D)]), nus=(D + 2.) * np.ones((K,)))
    npr.seed(seed + 1)
    comps = TiedGaussiansWithNormalWisharts(size=K, dim=D, prior=prior)
    m = BayesianMixtureOfGaussians(gating=gating, components=comps)
    put(out, "prior", nw_params(prior)); put(out, "gprior", gating_params(gating.prior, 'dirichlet'))
    npr.seed(seed + 2)
    m.resample(X.copy(), init_labels='random', maxiter=3, progress_bar=False)
    out["gibbs_mus"], out["gibbs_lmbdas"] = m.components.likelihood.mus, m.components.likelihood.lmbdas
    put(out, "gibbs_post", nw_params(m.components.posterior))
    out["gibbs_nat2"] = m.components.posterior.nat_param[2]       # recomputed from the pooled psi on read
    npr.seed(seed + 3)
    vlb = m.meanfield_coordinate_descent(X.copy(), randomize=False, maxiter=iters, tol=0., progress_bar=False)
    out["vi_vlb"] = np.array(vlb)
    put(out, "vi_post", nw_params(m.components.posterior))
    lik = MixtureOfGaussians(gating=Categorical(dim=K), components=TiedGaussiansWithPrecision(size=K, dim=D))
    npr.seed(seed + 4)
    out["em_loglik"] = np.array(lik.max_likelihood(X.copy(), randomize=True, maxiter=iters, progress_bar=False))
    out["em_mus"], out["em_lmbdas"] = lik.components.mus, lik.components.lmbdas
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok")


def tied_ilr_prediction_case(name, N, K, seed, gibbs_iters=4, svi_iters=5, vi_iters=5, dx=1, dy=1):
    """The flow of examples/ilr/evaluate_sine.py:88-155 at fixture size: scaled data, Normal-Wishart basis,
    TIED Matrix-Normal-Wishart models, stick-breaking gating; Gibbs -> SVI -> VI -> prior := posterior ->
    meanfield_prediction (ilr.py:325-430): average / mode, diagonal / full variance, Gaussian predictive."""
    import random
    from mimo.distributions import TiedMatrixNormalWisharts, TiedLinearGaussiansWithMatrixNormalWisharts
    npr.seed(seed)
    t = np.linspace(0., 4. * np.pi, N)
    if dx == 1:
        X = (t + 0.1 * npr.randn(N))[:, None]
    else:
        X = np.column_stack([t + 0.1 * npr.randn(N)] + [npr.randn(N) for _ in range(dx - 1)])
    Y = np.column_stack([3. * np.sin(t + 0.5 * j) + 0.3 * npr.randn(N) for j in range(dy)])
    if dx > 1:
        Y = Y + 0.5 * X[:, 1:2]
    train = np.r_[0:N // 4, N // 2:3 * N // 4]
    Xtr, Ytr = np.ascontiguousarray(X[train]), np.ascontiguousarray(Y[train])
    dc = dx + 1
    gating = make_gating(K, 'stick')
    bprior = StackedNormalWisharts(size=K, dim=dx, mus=np.zeros((K, dx)), kappas=1e-2 * np.ones((K,)),
                                   psis=np.stack(K * [1e2 * np.eye(dx)]), nus=(dx + 1.) * np.ones((K,)) + 1e-16)
    npr.seed(seed + 1)
    basis = StackedGaussiansWithNormalWisharts(size=K, dim=dx, prior=bprior)
    mprior = TiedMatrixNormalWisharts(K, dc, dy, Ms=np.zeros((K, dy, dc)), Ks=np.stack(K * [1e-2 * np.eye(dc)]),
                                      psis=np.stack(K * [1e1 * np.eye(dy)]), nus=(dy + 1.) * np.ones((K,)) + 1e-16)
    models = TiedLinearGaussiansWithMatrixNormalWisharts(K, dc, dy, mprior, affine=True)
    ilr = BayesianMixtureOfLinearGaussians(size=K, input_dim=dx, output_dim=dy, gating=gating, basis=basis, models=models)
    out = dict(X=X, Y=Y, Xtr=Xtr, Ytr=Ytr, gating_kind=np.array('stick'), K=np.array(K), seed=np.array(seed),
               gibbs_iters=np.array(gibbs_iters), svi_iters=np.array(svi_iters), vi_iters=np.array(vi_iters))
    put(out, "bprior", nw_params(ilr.basis.prior)); put(out, "mprior", mnw_params(ilr.models.prior))
    put(out, "gprior", gating_params(ilr.gating.prior, 'stick'))
    ilr.init_transform(Xtr, Ytr)
    npr.seed(seed + 2); random.seed(seed + 3)
    ilr.resample(Xtr.copy(), Ytr.copy(), init_labels='random', maxiter=gibbs_iters, progress_bar=False)
    out["gibbs_As"], out["gibbs_lmbdas"] = ilr.models.likelihood.As, ilr.models.likelihood.lmbdas
    put(out, "gibbs_mpost", mnw_params(ilr.models.posterior))
    out["svi_vlb"] = np.array(ilr.meanfield_stochastic_descent(Xtr.copy(), Ytr.copy(), randomize=False, maxiter=svi_iters,
                                                               step_size=5e-1, batch_size=64, progress_bar=False))
    put(out, "svi_mpost", mnw_params(ilr.models.posterior))
    out["vi_vlb"] = np.array(ilr.meanfield_coordinate_descent(Xtr.copy(), Ytr.copy(), randomize=False, maxiter=vi_iters,
                                                              tol=0., progress_bar=False))
    put(out, "vi_mpost", mnw_params(ilr.models.posterior)); put(out, "vi_bpost", nw_params(ilr.basis.posterior))
    put(out, "vi_gpost", gating_params(ilr.gating.posterior, 'stick'))
    ilr.basis.prior = ilr.basis.posterior
    ilr.models.prior = ilr.models.posterior
    # prediction on ALL inputs (train + held-out), every variant
    # dist='studentt' raises in the reference for stacked blocks (stats.py:79 divides (K,N) by (K,);
    # ilr.py:355 contracts a 4-D array with 'ndl') — only the Gaussian predictive has observable behaviour
    for dist in ('gaussian',):
        xx = ilr.input_transform.transform(X)
        out[f"pred_weights_{dist}"] = ilr.meanfield_predictive_weights(xx, dist)
        out[f"pred_activation_{dist}"] = ilr.meanfield_predictive_activation(X.copy(), dist)
        mus, covars = ilr.meanfield_predictive_moments(xx, dist)
        out[f"pred_mus_{dist}"], out[f"pred_covars_{dist}"] = mus, covars
        for pred in ('average', 'mode'):
            # (with y the reference raises: stacked_mvn_logpdf gets (K,N,d) means, stats.py:57 — no nlpd vector)
            mu, var, std = ilr.meanfield_prediction(X.copy(), prediction=pred, dist=dist)
            out[f"pred_{pred}_{dist}_mu"], out[f"pred_{pred}_{dist}_var"] = mu, var
            out[f"pred_{pred}_{dist}_std"] = std
    mu, covar, std = ilr.meanfield_prediction(X.copy(), prediction='average', variance='full')
    out["pred_average_gaussian_covar"] = covar
    out["basis_logpred_gaussian"] = ilr.basis.log_posterior_predictive_gaussian(ilr.input_transform.transform(X))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok")


def ng_params(p):
    return dict(mus=p.mus, kappas=p.kappas, alphas=p.alphas, betas=p.betas)


def diag_gmm_case(name, N, D, K, seed, tied, iters=6):
    """Diagonal-precision GMM under (Tied)NormalGammas (composite.py:286-547, bayesian.py:343-500,
    gaussian.py:697-878; examples/dgmm, examples/tdgmm): seeded Gibbs sweeps, VI / SVI / MAP / EM traces and the
    E-step tables.  The reference's stacked alpha / beta setters write unused attributes (composite.py:472-484),
    so what is pinned here is its OBSERVABLE behaviour: the posterior keeps the prior's Gamma factors."""
    from mimo.distributions import (StackedNormalGammas, TiedNormalGammas, StackedGaussiansWithNormalGammas,
                                    TiedGaussiansWithNormalGammas, StackedGaussiansWithDiagonalPrecision,
                                    TiedGaussiansWithDiagonalPrecision, Categorical)
    from mimo.mixtures import MixtureOfGaussians
    Prior = TiedNormalGammas if tied else StackedNormalGammas
    Comp = TiedGaussiansWithNormalGammas if tied else StackedGaussiansWithNormalGammas
    Lik = TiedGaussiansWithDiagonalPrecision if tied else StackedGaussiansWithDiagonalPrecision
    npr.seed(seed)
    X = make_data(N, D)
    out = dict(X=X, K=np.array(K), D=np.array(D), seed=np.array(seed), iters=np.array(iters), tied=np.array(tied),
               gating_kind=np.array('dirichlet'))
    gating = CategoricalWithDirichlet(dim=K, prior=Dirichlet(dim=K, alphas=2. * np.ones((K,))))   # MAP needs alpha > 1
    prior = Prior(size=K, dim=D, mus=np.zeros((K, D)), kappas=1e-2 * np.ones((K, D)),
                  alphas=(D + 1. + 1e-8) / 2. * np.ones((K, D)), betas=0.5 * np.ones((K, D)))
    npr.seed(seed + 1)
    comps = Comp(size=K, dim=D, prior=prior)
    m = BayesianMixtureOfGaussians(gating=gating, components=comps)
    put(out, "prior", ng_params(prior)); put(out, "gprior", gating_params(gating.prior, 'dirichlet'))
    out["init_mus"], out["init_lmbdas_diags"] = comps.likelihood.mus, comps.likelihood.lmbdas_diags
    out["loglik_table"] = comps.likelihood.log_likelihood(X.copy())
    npr.seed(seed + 2)
    m.resample(X.copy(), init_labels='random', maxiter=3, progress_bar=False)
    out["gibbs_mus"], out["gibbs_lmbdas_diags"] = m.components.likelihood.mus, m.components.likelihood.lmbdas_diags
    out["gibbs_probs"] = m.gating.likelihood.probs
    put(out, "gibbs_post", ng_params(m.components.posterior))
    npr.seed(seed + 3)
    vlb = m.meanfield_coordinate_descent(X.copy(), randomize=False, maxiter=iters, tol=0., progress_bar=False)
    out["vi_vlb"] = np.array(vlb)
    put(out, "vi_post", ng_params(m.components.posterior))
    out["vi_ell_table"] = m.components.expected_log_likelihood(X.copy())
    out["vi_resp"] = m.expected_responsibilities(X.copy())
    out["vi_nat"] = np.stack(m.components.posterior.nat_param)
    st = m.components.likelihood.weighted_statistics(X.copy(), out["vi_resp"])
    out["vi_stats_x"], out["vi_stats_nd"], out["vi_stats_xx"] = st[0], st[1], st[3]
    import random
    npr.seed(seed + 4); random.seed(seed + 14)
    out["svi_vlb"] = np.array(m.meanfield_stochastic_descent(X.copy(), randomize=False, maxiter=iters, step_size=5e-1,
                                                             batch_size=64, progress_bar=False))
    put(out, "svi_post", ng_params(m.components.posterior))
    npr.seed(seed + 5)
    out["map_logprob"] = np.array(m.max_aposteriori(X.copy(), randomize=True, maxiter=iters, progress_bar=False))
    out["map_mus"], out["map_lmbdas_diags"] = m.components.likelihood.mus, m.components.likelihood.lmbdas_diags
    lik = MixtureOfGaussians(gating=Categorical(dim=K), components=Lik(size=K, dim=D))
    npr.seed(seed + 6)
    out["em_loglik"] = np.array(lik.max_likelihood(X.copy(), randomize=True, maxiter=iters, progress_bar=False))
    out["em_mus"], out["em_lmbdas_diags"] = lik.components.mus, lik.components.lmbdas_diags
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok")


def hier_gmm_case(name, N, D, K, M, seed, iters=4, sub=3):
    """Hierarchical mixtures (hgmm.py:118-504, bayesian.py:592-793; examples/hgmm): K Gaussians with one precision
    under a Normal-Wishart hyper-prior — seeded Gibbs sweeps, VI traces without and with per-datum weights, the
    full-data natural-gradient driver and the E-step tables — and mixtures of M such mixtures (Gibbs, VI, SVI, EM)."""
    import random
    from mimo.distributions import (NormalWishart, TiedGaussiansWithScaledPrecision, TiedGaussiansWithPrecision,
                                    TiedGaussiansWithHierarchicalNormalWisharts, Categorical)
    from mimo.mixtures import (BayesianMixtureOfGaussiansWithHierarchicalPrior, BayesianMixtureOfMixtureOfGaussians,
                               MixtureOfGaussians)
    from mimo.mixtures.hgmm import MixtureOfMixtureOfGaussians
    npr.seed(seed)
    X = make_data(N, D, n_clusters=K)
    w = np.linspace(0.25, 1., N)
    out = dict(X=X, w=w, K=np.array(K), M=np.array(M), D=np.array(D), seed=np.array(seed), iters=np.array(iters),
               sub=np.array(sub))

    def inner(k):
        gating = CategoricalWithDirichlet(dim=k, prior=Dirichlet(dim=k, alphas=np.ones((k,))))
        hyper = NormalWishart(dim=D, mu=np.zeros((D,)), kappa=1e-2, psi=np.eye(D), nu=D + 1. + 1e-8)
        prior = TiedGaussiansWithScaledPrecision(size=k, dim=D, kappas=1e-2 * np.ones((k,)))
        comps = TiedGaussiansWithHierarchicalNormalWisharts(size=k, dim=D, hyper_prior=hyper, prior=prior)
        return BayesianMixtureOfGaussiansWithHierarchicalPrior(size=k, dim=D, gating=gating, components=comps)

    def state(m, pre):
        c = m.components
        out[pre + "_post_mus"], out[pre + "_post_kappas"] = c.posterior.mus, c.posterior.kappas
        out[pre + "_post_lmbdas"] = c.posterior.lmbdas
        for nm, v in zip(("mu", "kappa", "psi", "nu"), c.hyper_posterior.params):
            out[pre + "_hyper_" + nm] = np.asarray(v)
        out[pre + "_lik_mus"], out[pre + "_lik_lmbdas"] = c.likelihood.mus, c.likelihood.lmbdas
        out[pre + "_galphas"] = m.gating.posterior.alphas

    npr.seed(seed + 1); m = inner(K)
    out["init_lik_mus"], out["init_lik_lmbdas"] = m.components.likelihood.mus, m.components.likelihood.lmbdas
    npr.seed(seed + 2)
    m.resample(X.copy(), maxiter=iters, maxsubiter=sub, progress_bar=False)
    state(m, "gibbs")
    npr.seed(seed + 3)
    out["vi_vlb"] = np.array(m.meanfield_coordinate_descent(X.copy(), randomize=False, maxiter=iters, maxsubiter=sub, tol=0.,
                                                            progress_bar=False))
    state(m, "vi")
    out["vi_ell_table"] = m.components.expected_log_likelihood(X.copy())
    out["vi_resp"] = m.expected_responsibilities(X.copy())
    out["vi_comp_vlb"] = np.array(m.components.variational_lowerbound())
    out["vi_logpred"] = m.components.log_posterior_predictive_gaussian(X.copy())
    npr.seed(seed + 4)
    out["viw_vlb"] = np.array(m.meanfield_coordinate_descent(X.copy(), randomize=True, weights=w, maxiter=iters,
                                                             maxsubiter=sub, tol=0., progress_bar=False))
    state(m, "viw")
    npr.seed(seed + 5)
    m.meanfield_stochastic_descent(X.copy(), randomize=False, weights=w, maxiter=iters, maxsubiter=sub, step_size=5e-1,
                                   progress_bar=False)
    state(m, "svi")

    def outer():
        gating = CategoricalWithDirichlet(dim=M, prior=Dirichlet(dim=M, alphas=np.ones((M,))))
        return BayesianMixtureOfMixtureOfGaussians(cluster_size=M, mixture_size=K, dim=D, gating=gating,
                                                   components=[inner(K) for _ in range(M)])

    def mom_state(mm, pre):
        out[pre + "_galphas"] = mm.gating.posterior.alphas
        out[pre + "_post_mus"] = np.stack([c.components.posterior.mus for c in mm.components])
        out[pre + "_hyper_psi"] = np.stack([c.components.hyper_posterior.wishart.psi for c in mm.components])
        out[pre + "_inner_galphas"] = np.stack([c.gating.posterior.alphas for c in mm.components])

    npr.seed(seed + 6); mm = outer()
    npr.seed(seed + 7)
    mm.meanfield_coordinate_descent(X.copy(), randomize=True, maxiter=3, maxsubiter=2, maxsubsubiter=2, progress_bar=False)
    mom_state(mm, "mom_vi")
    out["mom_vi_resp"] = mm.expected_responsibilities(X.copy())
    npr.seed(seed + 8); random.seed(seed + 18)
    mm.meanfield_stochastic_descent(X.copy(), randomize=False, maxiter=3, maxsubiter=2, maxsubsubiter=2, step_size=5e-1,
                                    batch_size=64, progress_bar=False)
    mom_state(mm, "mom_svi")
    npr.seed(seed + 9)
    mm.resample(X.copy(), init_labels='random', maxiter=2, maxsubiter=2, maxsubsubiter=2, progress_bar=False)
    mom_state(mm, "mom_gibbs")
    out["mom_gibbs_lik_mus"] = np.stack([c.components.likelihood.mus for c in mm.components])
    npr.seed(seed + 10)
    comps = [MixtureOfGaussians(gating=Categorical(dim=K), components=TiedGaussiansWithPrecision(size=K, dim=D))
             for _ in range(M)]
    em = MixtureOfMixtureOfGaussians(cluster_size=M, mixture_size=K, dim=D, gating=Categorical(dim=M), components=comps)
    out["mom_em_loglik"] = np.array(em.max_likelihood(X.copy(), randomize=True, maxiter=3, maxsubiter=2, progress_bar=False))
    out["mom_em_mus"] = np.stack([c.components.mus for c in em.components])
    out["mom_em_probs"] = em.gating.probs
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok")


def hier_ilr_case(name, N, dx, dy, K, seed, iters=4, sub=3):
    """Tied-activation mixture of linear-Gaussian experts (hilr.py:79-290, bayesian.py:1222-1522; examples/hilr):
    shared slope and output precision, per-expert offsets, hierarchical input density — seeded Gibbs sweeps, VI
    without and with per-datum weights, tables and the bound for explicit responsibilities."""
    from mimo.distributions import (NormalWishart, TiedGaussiansWithScaledPrecision, TiedGaussiansWithHierarchicalNormalWisharts,
                                    MatrixNormalWithPrecision, Wishart, TiedAffineLinearGaussiansWithMatrixNormalWisharts)
    from mimo.mixtures.hilr import BayesianMixtureOfLinearGaussiansWithTiedActivation
    npr.seed(seed)
    X = np.sort(npr.uniform(-6., 6., size=(N, dx)), axis=0)
    A = npr.randn(dy, dx)
    seg = np.minimum((X[:, 0] + 6.) / 12. * K, K - 1).astype(int)
    offs = 4. * npr.randn(K, dy)
    Y = X @ A.T + offs[seg] + 0.3 * npr.randn(N, dy)
    w = np.linspace(0.25, 1., N)
    out = dict(X=X, Y=Y, w=w, K=np.array(K), dx=np.array(dx), dy=np.array(dy), seed=np.array(seed), iters=np.array(iters),
               sub=np.array(sub))

    def build():
        gating = CategoricalWithDirichlet(dim=K, prior=Dirichlet(dim=K, alphas=np.ones((K,))))
        bh = NormalWishart(dim=dx, mu=np.zeros((dx,)), kappa=1e-2, psi=np.eye(dx), nu=dx + 1. + 1e-8)
        bp = TiedGaussiansWithScaledPrecision(size=K, dim=dx, kappas=1e-2 * np.ones((K,)))
        basis = TiedGaussiansWithHierarchicalNormalWisharts(size=K, dim=dx, hyper_prior=bh, prior=bp)
        sp = MatrixNormalWithPrecision(column_dim=dx, row_dim=dy, M=np.zeros((dy, dx)), K=1e-2 * np.eye(dx))
        op = TiedGaussiansWithScaledPrecision(size=K, dim=dy, mus=np.zeros((K, dy)), kappas=1e-2 * np.ones((K,)))
        pp = Wishart(dim=dy, psi=np.eye(dy), nu=dy + 1. + 1e-16)
        models = TiedAffineLinearGaussiansWithMatrixNormalWisharts(size=K, column_dim=dx, row_dim=dy, slope_prior=sp,
                                                                   offset_prior=op, precision_prior=pp)
        return BayesianMixtureOfLinearGaussiansWithTiedActivation(size=K, input_dim=dx, output_dim=dy, gating=gating,
                                                                  basis=basis, models=models)

    def state(m, pre):
        mo, ba = m.models, m.basis
        out[pre + "_slope_M"], out[pre + "_slope_K"] = mo.slope_posterior.M, mo.slope_posterior.K
        out[pre + "_prec_psi"], out[pre + "_prec_nu"] = mo.precision_posterior.psi, np.asarray(mo.precision_posterior.nu)
        out[pre + "_off_mus"], out[pre + "_off_kappas"] = mo.offset_posterior.mus, mo.offset_posterior.kappas
        out[pre + "_off_lmbdas"] = mo.offset_posterior.lmbdas
        out[pre + "_lik_As"], out[pre + "_lik_cs"], out[pre + "_lik_lmbdas"] = mo.likelihood.As, mo.likelihood.cs, mo.likelihood.lmbdas
        out[pre + "_basis_mus"], out[pre + "_basis_hyper_psi"] = ba.posterior.mus, ba.hyper_posterior.wishart.psi
        out[pre + "_galphas"] = m.gating.posterior.alphas

    npr.seed(seed + 1); m = build()
    state(m, "init")
    out["init_loglik"] = m.models.likelihood.log_likelihood(X.copy(), Y.copy())
    npr.seed(seed + 2)
    m.resample(X.copy(), Y.copy(), maxiter=iters, maxsubiter=sub, progress_bar=False)
    state(m, "gibbs")
    npr.seed(seed + 3)
    out["vi_return"] = np.array(m.meanfield_coordinate_descent(X.copy(), Y.copy(), randomize=False, maxiter=iters, maxsubiter=sub,
                                                               progress_bar=False))
    state(m, "vi")
    out["vi_models_ell"] = m.models.expected_log_likelihood(X.copy(), Y.copy())
    out["vi_basis_ell"] = m.basis.expected_log_likelihood(X.copy())
    out["vi_resp"] = m.expected_responsibilities(X.copy(), Y.copy())
    out["vi_models_vlb"] = np.asarray(m.models.variational_lowerbound())
    out["vi_vlb"] = np.asarray(m.variational_lowerbound(X.copy(), Y.copy(), out["vi_resp"]))
    npr.seed(seed + 4)
    m.meanfield_coordinate_descent(X.copy(), Y.copy(), randomize=True, weights=w, maxiter=iters, maxsubiter=sub, progress_bar=False)
    state(m, "viw")
    try:
        m.meanfield_stochastic_descent(X.copy(), Y.copy(), randomize=False, maxiter=2, maxsubiter=2, progress_bar=False)
        out["svi_raises"] = np.array(False)
    except NotImplementedError:
        out["svi_raises"] = np.array(True)

    # mixture of M such mixtures (hilr.py:293-609): scaled data, VI, Gibbs, prediction
    from mimo.mixtures.hilr import BayesianMixtureOfMixtureOfLinearGaussians
    M = 2
    out["M"] = np.array(M)

    def mom_state(mm, pre):
        out[pre + "_galphas"] = mm.gating.posterior.alphas
        out[pre + "_slope_M"] = np.stack([c.models.slope_posterior.M for c in mm.components])
        out[pre + "_off_mus"] = np.stack([c.models.offset_posterior.mus for c in mm.components])
        out[pre + "_prec_psi"] = np.stack([c.models.precision_posterior.psi for c in mm.components])
        out[pre + "_basis_mus"] = np.stack([c.basis.posterior.mus for c in mm.components])
        out[pre + "_inner_galphas"] = np.stack([c.gating.posterior.alphas for c in mm.components])

    npr.seed(seed + 6)
    gating = CategoricalWithDirichlet(dim=M, prior=Dirichlet(dim=M, alphas=np.ones((M,))))
    mm = BayesianMixtureOfMixtureOfLinearGaussians(cluster_size=M, mixture_size=K, input_dim=dx, output_dim=dy, gating=gating,
                                                   components=[build() for _ in range(M)])
    mm.init_transform(X, Y)
    npr.seed(seed + 7)
    mm.meanfield_coordinate_descent(X.copy(), Y.copy(), randomize=True, maxiter=3, maxsubiter=2, maxsubsubiter=2, progress_bar=False)
    mom_state(mm, "mom_vi")
    xx = mm.input_transform.transform(X)
    out["mom_pred_weights"] = mm.meanfield_predictive_weights(xx)
    out["mom_pred_activation"] = mm.meanfield_predictive_activation(X.copy())
    mus, covars = mm.meanfield_predictive_moments(xx)
    out["mom_pred_mus"], out["mom_pred_covars"] = mus, covars
    for pred in ("average", "mode"):
        mu, var, std = mm.meanfield_prediction(X.copy(), prediction=pred)
        out[f"mom_pred_{pred}_mu"], out[f"mom_pred_{pred}_var"], out[f"mom_pred_{pred}_std"] = mu, var, std
    out["mom_pred_average_covar"] = mm.meanfield_prediction(X.copy(), prediction='average', variance='full')[1]
    npr.seed(seed + 8)
    mm.resample(X.copy(), Y.copy(), init_labels='random', maxiter=2, maxsubiter=2, maxsubsubiter=2, progress_bar=False)
    mom_state(mm, "mom_gibbs")
    try:
        mm.meanfield_stochastic_descent(X.copy(), Y.copy(), randomize=False, maxiter=1, maxsubiter=1, maxsubsubiter=1,
                                        batch_size=32, progress_bar=False)
        out["mom_svi_raises"] = np.array(False)
    except NotImplementedError:
        out["mom_svi_raises"] = np.array(True)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok")


def nan_rows_case(name, N, D, K, seed, n_bad=9):
    """Method-level behaviour of the reference on rows that hold a NaN (fresh copies per call: the reference edits
    the caller's array in place, gaussian.py:513): log-density tables, responsibilities, statistics, counts."""
    npr.seed(seed)
    X = make_data(N, D)
    bad = np.sort(npr.choice(N, size=n_bad, replace=False))
    Xn = X.copy()
    for i, r in enumerate(bad):
        Xn[r, i % D] = np.nan
        if i % 3 == 0:
            Xn[r, :] = np.nan
    gating = make_gating(K, 'dirichlet')
    prior = StackedNormalWisharts(size=K, dim=D, mus=np.zeros((K, D)), kappas=1e-2 * np.ones((K,)),
                                  psis=np.stack(K * [np.eye(D)]), nus=(D + 1.) * np.ones((K,)) + 1e-8)
    comps = StackedGaussiansWithNormalWisharts(size=K, dim=D, prior=prior)
    model = BayesianMixtureOfGaussians(gating=gating, components=comps)
    resp0 = npr.rand(K, N)
    resp0 /= np.sum(resp0, axis=0)
    model.meanfield_update_parameters(X.copy(), resp0)            # a sensible point estimate (from the complete data)
    out = dict(X=Xn, bad=bad, K=np.array(K), D=np.array(D), resp0=resp0,
               lik_mus=model.components.likelihood.mus, lik_lmbdas=model.components.likelihood.lmbdas,
               lik_probs=model.gating.likelihood.probs)
    out["A1_loglik"] = model.components.likelihood.log_likelihood(Xn.copy())
    out["A2_lcl"] = model.likelihood.log_complete_likelihood(Xn.copy())
    out["A2_resp"] = model.likelihood.responsibilities(Xn.copy())
    out["A2_ll"] = model.likelihood.log_likelihood(Xn.copy())
    st = model.components.likelihood.weighted_statistics(Xn.copy(), out["A2_resp"])
    put(out, "stats", dict(xk=st[0], nk=st[1], xxTk=st[2]))
    out["counts"] = model.gating.likelihood.weighted_statistics(None, out["A2_resp"])
    st0 = model.components.likelihood.weighted_statistics(Xn.copy(), resp0)
    put(out, "stats0", dict(xk=st0[0], nk=st0[1], xxTk=st0[2]))
    u = npr.random(size=(1, N))
    with FixedUniforms(u):
        _, labels = model.resample_labels(Xn.copy())
    out["u"], out["labels"] = u, labels
    ls = model.components.likelihood.weighted_statistics(Xn.copy(), one_hot(labels, K))
    put(out, "lstats", dict(xk=ls[0], nk=ls[1], xxTk=ls[2]))
    out["lcounts"] = model.gating.likelihood.statistics(labels)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok")


def nan_rows_ilr_case(name, N, dx, dy, K, seed):
    """Method-level behaviour of the reference's linear-Gaussian mixture on rows that hold a NaN in x, in y, or in both
    (fresh copies per call): lingauss.py:150-151 zeroes the data part of a row only when x AND y hold a NaN and evaluates
    the other rows on nan_to_num'ed values; inside log_complete_likelihood (ilr.py:71-75) the input density has already
    nan_to_num'ed x IN PLACE when the experts' density runs, so there no row is zeroed at all; the statistics drop every row
    with a NaN in x or y (lingauss.py:103-104, 306-310)."""
    npr.seed(seed)
    X = make_data(N, dx)
    Atrue = npr.randn(4, dy, dx)
    Y = np.ascontiguousarray(np.einsum('ndl,nl->nd', Atrue[npr.randint(4, size=N)], X) + 0.3 * npr.randn(N, dy))
    dc = dx + 1
    gating = make_gating(K, 'dirichlet')
    bprior = StackedNormalWisharts(size=K, dim=dx, mus=np.zeros((K, dx)), kappas=1e-2 * np.ones((K,)),
                                   psis=np.stack(K * [1e2 * np.eye(dx)]), nus=(dx + 1.) * np.ones((K,)) + 1e-16)
    basis = StackedGaussiansWithNormalWisharts(size=K, dim=dx, prior=bprior)
    mprior = StackedMatrixNormalWisharts(K, dc, dy, Ms=np.zeros((K, dy, dc)), Ks=np.stack(K * [1e-2 * np.eye(dc)]),
                                         psis=np.stack(K * [np.eye(dy)]), nus=(dy + 1.) * np.ones((K,)) + 1e-16)
    models = StackedLinearGaussiansWithMatrixNormalWisharts(K, dc, dy, mprior, affine=True)
    ilr = BayesianMixtureOfLinearGaussians(size=K, input_dim=dx, output_dim=dy, gating=gating, basis=basis, models=models)
    resp0 = npr.rand(K, N)
    resp0 /= np.sum(resp0, axis=0)
    ilr.meanfield_update_parameters(X.copy(), Y.copy(), resp0)        # a sensible point estimate (from the complete data)
    bad = np.sort(npr.choice(N, size=12, replace=False))
    Xn, Yn = X.copy(), Y.copy()
    for i, r in enumerate(bad):
        kind = i % 3                       # 0: NaN in x only, 1: in y only, 2: in both
        if kind in (0, 2):
            Xn[r, i % dx] = np.nan         # (one element: the others of the row keep their values under nan_to_num)
        if kind in (1, 2):
            Yn[r, i % dy] = np.nan
    out = dict(X=Xn, Y=Yn, bad=bad, K=np.array(K), resp0=resp0)
    put(out, "lik", dict(mus=ilr.basis.likelihood.mus, lmbdas=ilr.basis.likelihood.lmbdas,
                         As=ilr.models.likelihood.As, lmbdas_y=ilr.models.likelihood.lmbdas,
                         probs=ilr.gating.likelihood.probs))
    out["A1_basis_loglik"] = ilr.basis.likelihood.log_likelihood(Xn.copy())
    out["A5_loglik"] = ilr.models.likelihood.log_likelihood(Xn.copy(), Yn.copy())
    out["A7_lcl"] = ilr.likelihood.log_complete_likelihood(Xn.copy(), Yn.copy())
    out["A7_resp"] = ilr.likelihood.responsibilities(Xn.copy(), Yn.copy())
    out["A7_ll"] = ilr.likelihood.log_likelihood(Xn.copy(), Yn.copy())
    ms = ilr.models.likelihood.weighted_statistics(Xn.copy(), Yn.copy(), resp0)
    put(out, "mstats0", dict(yxTk=ms[0], xxTk=ms[1], yyTk=ms[2], nk=ms[3]))
    bs = ilr.basis.likelihood.weighted_statistics(Xn.copy(), resp0)
    put(out, "bstats0", dict(xk=bs[0], nk=bs[1], xxTk=bs[2]))
    u = npr.random(size=(1, N))
    with FixedUniforms(u):
        _, labels = ilr.resample_labels(Xn.copy(), Yn.copy())
    out["u"], out["labels"] = u, labels
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "nan_ilr":
        nan_rows_ilr_case("nan_rows_ilr_dx2_dy1_k6", N=300, dx=2, dy=1, K=6, seed=1363)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "n4099":
        # N = 4099 per config (tile tails across many workgroups against REFERENCE output), K capped
        gmm_case("gmm_c2_d16_k16_n4099", N=4099, D=16, K=16, kind='dirichlet', seed=1364, vi_iters=3)
        gmm_case("gmm_c3_d8_k32_n4099", N=4099, D=8, K=32, kind='stick', seed=1365, vi_iters=3)
        ilr_case("ilr_c4_dx8_dy4_k16_n4099", N=4099, dx=8, dy=4, K=16, kind='stick', seed=1366, vi_iters=3)
        ilr_case("ilr_dx1_dy1_k50_stick", N=257, dx=1, dy=1, K=50, kind='stick', seed=1367, vi_iters=5)
        ilr_case("ilr_dx1_dy1_k50_n4099", N=4099, dx=1, dy=1, K=50, kind='stick', seed=1368, vi_iters=3)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "nan":
        nan_rows_case("nan_rows_gmm_d3_k5", N=400, D=3, K=5, seed=1361)
        nan_rows_case("nan_rows_gmm_d16_k70", N=300, D=16, K=70, seed=1362, n_bad=20)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "hilr":
        hier_ilr_case("hier_ilr_dx1_dy1_k3", N=300, dx=1, dy=1, K=3, seed=1356)
        hier_ilr_case("hier_ilr_dx2_dy2_k4", N=400, dx=2, dy=2, K=4, seed=1357)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "hier":
        hier_gmm_case("hier_gmm_d2_k4_m2", N=400, D=2, K=4, M=2, seed=1354)
        hier_gmm_case("hier_gmm_d3_k3_m3", N=300, D=3, K=3, M=3, seed=1355)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "diag":
        diag_gmm_case("diag_gmm_d3_k5", N=500, D=3, K=5, seed=1352, tied=False)
        diag_gmm_case("tied_diag_gmm_d4_k6", N=400, D=4, K=6, seed=1353, tied=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "tied":
        tied_gmm_case("tied_gmm_d3_k5", N=500, D=3, K=5, seed=1349)
        tied_ilr_prediction_case("tied_ilr_sine_k8", N=400, K=8, seed=1350)
        tied_ilr_prediction_case("tied_ilr_dx3_dy2_k6", N=300, K=6, seed=1351, dx=3, dy=2)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "fullk":
        # one full-K, N = 257 (129 at D = 32) fixture per BASELINE config whose K was capped above
        gmm_case("gmm_c3_d8_k256_stick", N=257, D=8, K=256, kind='stick', seed=1358, vi_iters=5)
        ilr_case("ilr_c4_dx8_dy4_k64_stick", N=257, dx=8, dy=4, K=64, kind='stick', seed=1359, vi_iters=5)
        gmm_case("gmm_c5_d32_k128_dir", N=129, D=32, K=128, kind='dirichlet', seed=1360, vi_iters=3)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "drivers":
        driver_traces_case("drivers_d3_k5_dir", N=600, D=3, K=5, kind='dirichlet', seed=1347)
        ilr_svi_case("ilr_svi_dx2_dy1_k8", N=500, dx=2, dy=1, K=8, seed=1348)
        sys.exit(0)
    # fixture-size versions of BASELINE.json configs C1..C5 (true D; K capped for file size, one full-K case)
    gmm_case("gmm_c1_d2_k4_dir", N=257, D=2, K=4, kind='dirichlet', seed=1337)
    gmm_case("gmm_c2_d16_k16_dir", N=257, D=16, K=16, kind='dirichlet', seed=1338)
    gmm_case("gmm_c2_d16_k64_dir", N=129, D=16, K=64, kind='dirichlet', seed=1339, vi_iters=5)
    gmm_case("gmm_c3_d8_k32_stick", N=257, D=8, K=32, kind='stick', seed=1340)
    gmm_case("gmm_c5_d32_k16_dir", N=129, D=32, K=16, kind='dirichlet', seed=1341, vi_iters=5)
    gmm_case("gmm_tail_d5_k7_stick", N=1031, D=5, K=7, kind='stick', seed=1342, vi_iters=10)
    ilr_case("ilr_c4_dx8_dy4_k16_stick", N=257, dx=8, dy=4, K=16, kind='stick', seed=1343)
    ilr_case("ilr_dx1_dy1_k6_dir", N=300, dx=1, dy=1, K=6, kind='dirichlet', seed=1344)
    gibbs_trace_case("gibbs_c1_trace", N=500, D=2, K=4, kind='dirichlet', seed=1345)
    gibbs_trace_case("gibbs_stick_trace", N=400, D=3, K=6, kind='stick', seed=1346)
    driver_traces_case("drivers_d3_k5_dir", N=600, D=3, K=5, kind='dirichlet', seed=1347)
    ilr_svi_case("ilr_svi_dx2_dy1_k8", N=500, dx=2, dy=1, K=8, seed=1348)
    tied_gmm_case("tied_gmm_d3_k5", N=500, D=3, K=5, seed=1349)
    tied_ilr_prediction_case("tied_ilr_sine_k8", N=400, K=8, seed=1350)
    tied_ilr_prediction_case("tied_ilr_dx3_dy2_k6", N=300, K=6, seed=1351, dx=3, dy=2)
    diag_gmm_case("diag_gmm_d3_k5", N=500, D=3, K=5, seed=1352, tied=False)
    diag_gmm_case("tied_diag_gmm_d4_k6", N=400, D=4, K=6, seed=1353, tied=True)
    hier_gmm_case("hier_gmm_d2_k4_m2", N=400, D=2, K=4, M=2, seed=1354)
    hier_gmm_case("hier_gmm_d3_k3_m3", N=300, D=3, K=3, M=3, seed=1355)
    hier_ilr_case("hier_ilr_dx1_dy1_k3", N=300, dx=1, dy=1, K=3, seed=1356)
    hier_ilr_case("hier_ilr_dx2_dy2_k4", N=400, dx=2, dy=2, K=4, seed=1357)
