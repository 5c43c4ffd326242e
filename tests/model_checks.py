"""Shared parity checks of the host API (mimo_amd.distributions / mimo_amd.mixtures) against the golden
vectors of the reference.  Run on CPU with the oracle-backed test double (tests/test_host_models.py)
and on the GPU with the real HipEngine through the C ABI (tests/test_gpu_parity.py)."""
import numpy as np
import pytest
import numpy.random as npr

from conftest import load_golden, rel_err, gating_of, nw_of, mnw_of
from mimo_amd.distributions import (Dirichlet, TruncatedStickBreaking, CategoricalWithDirichlet,
                                    CategoricalWithStickBreaking, StackedNormalWisharts,
                                    StackedGaussiansWithNormalWisharts, StackedMatrixNormalWisharts,
                                    StackedLinearGaussiansWithMatrixNormalWisharts, TiedNormalWisharts,
                                    TiedGaussiansWithNormalWisharts, TiedGaussiansWithPrecision,
                                    TiedMatrixNormalWisharts, TiedLinearGaussiansWithMatrixNormalWisharts,
                                    StackedNormalGammas, TiedNormalGammas, StackedGaussiansWithNormalGammas,
                                    TiedGaussiansWithNormalGammas, StackedGaussiansWithDiagonalPrecision,
                                    TiedGaussiansWithDiagonalPrecision)
from mimo_amd.mixtures import BayesianMixtureOfGaussians, BayesianMixtureOfLinearGaussians


def make_gating(g, K):
    kind, gprior = gating_of(g, "gprior")
    if kind == "dirichlet":
        return kind, CategoricalWithDirichlet(dim=K, prior=Dirichlet(dim=K, alphas=gprior.copy()))
    prior = TruncatedStickBreaking(dim=K, gammas=gprior[0].copy(), deltas=gprior[1].copy())
    return kind, CategoricalWithStickBreaking(dim=K, prior=prior)


def set_gating_posterior(gating, kind, params):
    if kind == "dirichlet":
        gating.posterior.alphas = params.copy()
    else:
        gating.posterior.gammas, gating.posterior.deltas = params[0].copy(), params[1].copy()


def gating_posterior(gating, kind):
    return gating.posterior.alphas if kind == "dirichlet" else (gating.posterior.gammas, gating.posterior.deltas)


def build_gmm(g, engine):
    K, D = int(g["K"]), int(g["D"])
    kind, gating = make_gating(g, K)
    prior = StackedNormalWisharts(size=K, dim=D, **{k: g["prior_" + k] for k in ("mus", "kappas", "psis", "nus")})
    comps = StackedGaussiansWithNormalWisharts(size=K, dim=D, prior=prior, engine=engine)
    return kind, BayesianMixtureOfGaussians(gating=gating, components=comps, engine=engine)


def load_gmm_state(model, g, kind, prefix="post", gprefix="gpost"):
    model.components.posterior.params = nw_of(g, prefix)
    set_gating_posterior(model.gating, kind, gating_of(g, gprefix)[1])
    if "lik_mus" in g:
        model.components.likelihood.params = (g["lik_mus"], g["lik_lmbdas"])
        model.gating.likelihood.params = g["lik_probs"].copy()


def check_gmm_case(name, engine, tol=1e-9):
    g = load_golden(name)
    X, K = g["X"], int(g["K"])
    kind, model = build_gmm(g, engine)
    load_gmm_state(model, g, kind)

    # Gibbs / EM form (A1, A2)
    assert rel_err(model.components.likelihood.log_likelihood(X), g["A1_loglik"]) < tol
    assert rel_err(model.likelihood.log_complete_likelihood(X), g["A2_lcl"]) < tol
    assert rel_err(model.likelihood.responsibilities(X), g["A2_resp"]) < tol
    assert rel_err(model.likelihood.log_likelihood(X), g["A2_ll"]) < tol
    # mean-field form (A3, A4)
    assert rel_err(model.components.expected_log_likelihood(X), g["A3_eloglik"]) < tol
    assert rel_err(model.expected_log_complete_likelihood(X), g["A4_elcl"]) < tol
    assert rel_err(model.expected_responsibilities(X), g["A4_eresp"]) < tol
    assert rel_err(model.expected_log_likelihood(X), g["A4_ell"]) < tol
    # statistics (A10, A12) for reference responsibilities and for random weights
    st = model.components.likelihood.weighted_statistics(X, g["A4_eresp"])
    assert rel_err(st[0], g["stats_xk"]) < tol and rel_err(st[1], g["stats_nk"]) < tol
    assert rel_err(st[2], g["stats_xxTk"]) < tol and rel_err(st[3], g["counts"]) < tol
    st0 = model.components.likelihood.weighted_statistics(X, g["resp0"])
    assert rel_err(st0[2], g["stats0_xxTk"]) < tol and rel_err(st0[0], g["stats0_xk"]) < tol
    # label draw (A8): reference uniforms and the engine's own counter-based stream — bit exact
    eng = model._bind(X)
    c, b, W = model.likelihood.canonical()
    labels, S = eng.gibbs_labels(c, b, W, u=g["u_mt"])
    assert labels.dtype == np.int32 and np.array_equal(labels, g["labels_mt"])
    assert rel_err(S.sxx, g["lstats_xxTk"]) < tol and np.array_equal(S.n, g["lcounts"])
    labels_p, _ = eng.gibbs_labels(c, b, W, seed=1337, sweep=3, stats=False)
    assert np.array_equal(labels_p, g["labels_philox"])
    S2 = eng.label_stats(g["labels_mt"], K)
    assert rel_err(S2.sxx, g["lstats_xxTk"]) < tol and rel_err(S2.sx, g["lstats_xk"]) < tol
    # ELBO pieces (A14)
    assert abs(model.variational_lowerbound_obs(X, g["A4_eresp"]) - g["vlb_obs"]) < tol * abs(g["vlb_obs"])
    assert abs(model.variational_lowerbound_labels(g["A4_eresp"]) - g["vlb_labels"]) < tol * abs(g["vlb_labels"])
    assert abs(model.gating.variational_lowerbound() - g["vlb_gating"]) < tol * max(1., abs(g["vlb_gating"]))
    assert rel_err(model.components.variational_lowerbound(), g["vlb_comps"]) < tol
    assert abs(model.variational_lowerbound(X, g["A4_eresp"]) - g["vlb_total"]) < tol * abs(g["vlb_total"])
    # the fused pass returns the same ELBO data terms and statistics in one go
    Sf, sc = eng.estep(*model.canonical_expected())
    assert abs(sc[0] - (g["vlb_obs"] + g["vlb_labels"])) < tol * abs(sc[0])
    assert rel_err(Sf.sxx, g["stats_xxTk"]) < tol and rel_err(Sf.n, g["counts"]) < tol
    # conjugate update (A13)
    model.meanfield_update_parameters(X, g["A4_eresp"])
    for a, bb in zip(model.components.posterior.params, nw_of(g, "post2")):
        assert rel_err(a, bb) < 1e-8
    for a, bb in zip(np.atleast_2d(gating_posterior(model.gating, kind)), np.atleast_2d(gating_of(g, "gpost2")[1])):
        assert rel_err(a, bb) < tol


def check_gmm_vi_trace(name, engine, tol=1e-8):
    g = load_golden(name)
    kind, model = build_gmm(g, engine)
    load_gmm_state(model, g, kind)
    vlb = model.meanfield_coordinate_descent(g["X"], randomize=False, maxiter=len(g["vi_vlb"]), tol=0.,
                                             progress_bar=False)
    assert rel_err(np.array(vlb), g["vi_vlb"]) < tol
    for a, b in zip(model.components.posterior.params, nw_of(g, "vi_post")):
        assert rel_err(a, b) < 1e-6
    assert np.all(np.diff(vlb) >= -1e-8 * abs(vlb[-1]))        # ELBO monotone (examples/gmm/toy/vi_toy.py:60)


def check_gibbs_trace(name, engine):
    """Seeded Gibbs run through the public driver reproduces the reference's labels and parameters."""
    g = load_golden(name)
    X, K = g["X"], int(g["K"])
    kind, model = build_gmm(g, engine)
    model.components.likelihood.params = (g["lik0_mus"], g["lik0_lmbdas"])
    model.gating.likelihood.params = g["lik0_probs"].copy()
    # explicit sweeps through the reference-shaped methods
    npr.seed(int(g["seed2"]))
    labels = model.gating.likelihood.rvs(len(X))
    assert np.array_equal(labels, g["labels_init"])
    for s in range(5):
        model.resample_components(X, labels)
        model.resample_gating(labels)
        _, labels = model.resample_labels(X)
        assert np.array_equal(labels, g[f"s{s}_labels"]), f"sweep {s}"
        assert rel_err(model.components.likelihood.mus, g[f"s{s}_mus"]) < 1e-8
        assert rel_err(model.gating.likelihood.probs, g[f"s{s}_probs"]) < 1e-10
    # the fused driver, same seed
    kind, model = build_gmm(g, engine)
    model.components.likelihood.params = (g["lik0_mus"], g["lik0_lmbdas"])
    model.gating.likelihood.params = g["lik0_probs"].copy()
    npr.seed(int(g["seed2"]))
    model.resample(X, init_labels='prior', maxiter=5, progress_bar=False, label_rng='host')
    assert np.array_equal(model.labels_, g["s4_labels"])
    assert rel_err(model.components.likelihood.mus, g["driver_mus"]) < 1e-8
    assert rel_err(model.gating.likelihood.probs, g["driver_probs"]) < 1e-10


def build_ilr(g, engine):
    K = int(g["K"])
    dx, dy = g["X"].shape[1], g["Y"].shape[1]
    kind, gating = make_gating(g, K)
    bprior = StackedNormalWisharts(size=K, dim=dx, **{k: g["bprior_" + k] for k in ("mus", "kappas", "psis", "nus")})
    basis = StackedGaussiansWithNormalWisharts(size=K, dim=dx, prior=bprior, engine=engine)
    mprior = StackedMatrixNormalWisharts(K, dx + 1, dy, **{k: g["mprior_" + k] for k in ("Ms", "Ks", "psis", "nus")})
    models = StackedLinearGaussiansWithMatrixNormalWisharts(K, dx + 1, dy, mprior, affine=True, engine=engine)
    ilr = BayesianMixtureOfLinearGaussians(size=K, input_dim=dx, output_dim=dy, gating=gating, basis=basis,
                                           models=models, engine=engine)
    return kind, ilr


def load_ilr_state(ilr, g, kind):
    ilr.basis.posterior.params = nw_of(g, "bpost")
    ilr.models.posterior.params = mnw_of(g, "mpost")
    set_gating_posterior(ilr.gating, kind, gating_of(g, "gpost")[1])
    ilr.basis.likelihood.params = (g["lik_mus"], g["lik_lmbdas"])
    ilr.models.likelihood.params = (g["lik_As"], g["lik_lmbdas_y"])
    ilr.gating.likelihood.params = g["lik_probs"].copy()


def check_ilr_case(name, engine, tol=1e-9):
    g = load_golden(name)
    X, Y, K = g["X"], g["Y"], int(g["K"])
    kind, ilr = build_ilr(g, engine)
    load_ilr_state(ilr, g, kind)
    assert rel_err(ilr.models.likelihood.log_likelihood(X, Y), g["A5_loglik"]) < tol
    assert rel_err(ilr.likelihood.log_complete_likelihood(X, Y), g["A7_lcl"]) < tol
    assert rel_err(ilr.likelihood.responsibilities(X, Y), g["A7_resp"]) < tol
    assert rel_err(ilr.models.expected_log_likelihood(X, Y), g["A6_eloglik"]) < tol
    assert rel_err(ilr.basis.expected_log_likelihood(X), g["A3_basis_eloglik"]) < tol
    assert rel_err(ilr.expected_log_complete_likelihood(X, Y), g["A7_elcl"]) < tol
    assert rel_err(ilr.expected_responsibilities(X, Y), g["A7_eresp"]) < tol
    ms = ilr.models.likelihood.weighted_statistics(X, Y, g["A7_eresp"])
    for a, b in zip(ms, (g["mstats_yxTk"], g["mstats_xxTk"], g["mstats_yyTk"], g["mstats_nk"])):
        assert rel_err(a, b) < tol
    bs = ilr.basis.likelihood.weighted_statistics(X, g["A7_eresp"])
    assert rel_err(bs[2], g["bstats_xxTk"]) < tol and rel_err(bs[0], g["bstats_xk"]) < tol
    eng = ilr._bind(X, Y)
    c, b, W = ilr.likelihood.canonical()
    labels, _ = eng.gibbs_labels(c, b, W, u=g["u_mt"], stats=False)
    assert np.array_equal(labels, g["labels_mt"])
    labels_p, _ = eng.gibbs_labels(c, b, W, seed=1337, sweep=3, stats=False)
    assert np.array_equal(labels_p, g["labels_philox"])
    assert abs(ilr.variational_lowerbound_data(X, Y, g["A7_eresp"]) - g["vlb_data"]) < tol * abs(g["vlb_data"])
    assert abs(ilr.variational_lowerbound_labels(g["A7_eresp"]) - g["vlb_labels"]) < tol * abs(g["vlb_labels"])
    assert rel_err(ilr.basis.variational_lowerbound(), g["vlb_basis"]) < tol
    assert rel_err(ilr.models.variational_lowerbound(), g["vlb_models"]) < 1e-8
    assert abs(ilr.variational_lowerbound(X, Y, g["A7_eresp"]) - g["vlb_total"]) < 1e-8 * abs(g["vlb_total"])
    ilr.meanfield_update_parameters(X, Y, g["A7_eresp"])
    for a, bb in zip(ilr.models.posterior.params, mnw_of(g, "mpost2")):
        assert rel_err(a, bb) < 1e-7
    for a, bb in zip(ilr.basis.posterior.params, nw_of(g, "bpost2")):
        assert rel_err(a, bb) < 1e-7


def check_ilr_vi_trace(name, engine, tol=1e-7):
    g = load_golden(name)
    kind, ilr = build_ilr(g, engine)
    load_ilr_state(ilr, g, kind)
    vlb = ilr.meanfield_coordinate_descent(g["X"], g["Y"], randomize=False, maxiter=len(g["vi_vlb"]), tol=0.,
                                           progress_bar=False)
    assert rel_err(np.array(vlb), g["vi_vlb"]) < tol
    for a, b in zip(ilr.models.posterior.params, mnw_of(g, "vi_mpost")):
        assert rel_err(a, b) < 1e-5


def check_driver_traces(name, engine, tol=1e-8):
    """EM / MAP / SVI drivers against seeded traces of the reference (gmm.py:77-103,176-204,300-326)."""
    import random
    from mimo_amd.distributions import Categorical, StackedGaussiansWithPrecision
    from mimo_amd.mixtures import MixtureOfGaussians
    g = load_golden(name)
    X, K, D, seed, iters = g["X"], int(g["K"]), int(g["D"]), int(g["seed"]), int(g["iters"])
    lik = MixtureOfGaussians(gating=Categorical(dim=K), components=StackedGaussiansWithPrecision(K, D, engine=engine),
                             engine=engine)
    npr.seed(seed + 10)
    ll = lik.max_likelihood(X, randomize=True, maxiter=iters, progress_bar=False)
    assert rel_err(np.array(ll), g["em_loglik"]) < tol
    assert rel_err(lik.components.mus, g["em_mus"]) < 1e-6 and rel_err(lik.gating.probs, g["em_probs"]) < 1e-6
    assert np.all(np.diff(ll) >= -1e-8 * abs(ll[-1]))          # EM log-likelihood monotone (em_toy.py:47)

    def fresh():
        kind, gating = make_gating(g, K)
        prior = StackedNormalWisharts(size=K, dim=D, **{k: g["prior_" + k] for k in ("mus", "kappas", "psis", "nus")})
        comps = StackedGaussiansWithNormalWisharts(size=K, dim=D, prior=prior, engine=engine)
        return BayesianMixtureOfGaussians(gating=gating, components=comps, engine=engine)
    m = fresh()
    npr.seed(seed + 21)
    lp = m.max_aposteriori(X, randomize=True, maxiter=iters, progress_bar=False)
    assert rel_err(np.array(lp), g["map_logprob"]) < tol
    assert rel_err(m.components.likelihood.mus, g["map_mus"]) < 1e-6

    m = fresh()
    npr.seed(seed + 31)
    random.seed(seed + 32)
    vlb = m.meanfield_stochastic_descent(X, randomize=True, maxiter=iters, step_size=5e-2, batch_size=64,
                                         progress_bar=False)
    assert rel_err(np.array(vlb), g["svi_vlb"]) < tol
    for a, b in zip(m.components.posterior.params, nw_of(g, "svi_post")):
        assert rel_err(a, b) < 1e-7


def check_ilr_svi(name, engine, tol=1e-7):
    """ILR: scaling transform + seeded Gibbs warm-up + SVI, as examples/ilr/evaluate_*.py run it."""
    import random
    g = load_golden(name)
    kind, ilr = build_ilr(g, engine)
    ilr.init_transform(g["X"], g["Y"])
    npr.seed(int(g["seed"]) + 1)
    random.seed(int(g["seed"]) + 2)
    ilr.resample(g["X"], g["Y"], init_labels='random', maxiter=3, progress_bar=False)
    assert rel_err(ilr.gating.likelihood.probs, g["gibbs_probs"]) < 1e-9
    assert rel_err(ilr.models.likelihood.As, g["gibbs_As"]) < 1e-7
    vlb = ilr.meanfield_stochastic_descent(g["X"], g["Y"], randomize=False, maxiter=int(g["iters"]), step_size=5e-1,
                                           batch_size=64, progress_bar=False)
    assert rel_err(np.array(vlb), g["svi_vlb"]) < tol
    for a, b in zip(ilr.models.posterior.params, mnw_of(g, "svi_mpost")):
        assert rel_err(a, b) < 1e-6


def check_tied_gmm(name, engine, tol=1e-8):
    """Tied-covariance GMM: seeded Gibbs sweeps, VI trace, EM trace."""
    from mimo_amd.distributions import Categorical
    from mimo_amd.mixtures import MixtureOfGaussians
    g = load_golden(name)
    X, K, D, seed, iters = g["X"], int(g["K"]), int(g["D"]), int(g["seed"]), int(g["iters"])
    kind, gating = make_gating(g, K)
    prior = TiedNormalWisharts(size=K, dim=D, **{k: g["prior_" + k] for k in ("mus", "kappas", "psis", "nus")})
    npr.seed(seed + 1)
    comps = TiedGaussiansWithNormalWisharts(size=K, dim=D, prior=prior, engine=engine)
    m = BayesianMixtureOfGaussians(gating=gating, components=comps, engine=engine)
    npr.seed(seed + 2)
    m.resample(X, init_labels='random', maxiter=3, progress_bar=False, label_rng='host')
    assert rel_err(m.components.likelihood.mus, g["gibbs_mus"]) < tol
    assert rel_err(m.components.likelihood.lmbdas, g["gibbs_lmbdas"]) < tol
    for a, b in zip(m.components.posterior.params, nw_of(g, "gibbs_post")):
        assert rel_err(a, b) < tol
    assert np.allclose(m.components.posterior.psis, m.components.posterior.psis[0])       # one shared psi
    assert rel_err(m.components.posterior.nat_param[2], g["gibbs_nat2"]) < tol           # recomputed on read
    npr.seed(seed + 3)
    vlb = m.meanfield_coordinate_descent(X, randomize=False, maxiter=iters, tol=0., progress_bar=False)
    assert rel_err(np.array(vlb), g["vi_vlb"]) < tol
    for a, b in zip(m.components.posterior.params, nw_of(g, "vi_post")):
        assert rel_err(a, b) < 1e-6
    lik = MixtureOfGaussians(gating=Categorical(dim=K), components=TiedGaussiansWithPrecision(K, D, engine=engine),
                             engine=engine)
    npr.seed(seed + 4)
    ll = lik.max_likelihood(X, randomize=True, maxiter=iters, progress_bar=False)
    assert rel_err(np.array(ll), g["em_loglik"]) < tol
    assert rel_err(lik.components.mus, g["em_mus"]) < 1e-6 and rel_err(lik.components.lmbdas, g["em_lmbdas"]) < 1e-6


def check_diag_gmm(name, engine, tol=1e-8):
    """Diagonal-precision GMM under (Tied)NormalGammas (examples/dgmm, examples/tdgmm):
    tables, seeded Gibbs sweeps, VI / SVI / MAP / EM traces against the reference's observable behaviour."""
    from mimo_amd.distributions import Categorical
    from mimo_amd.mixtures import MixtureOfGaussians
    g = load_golden(name)
    X, K, D, seed, iters, tied = g["X"], int(g["K"]), int(g["D"]), int(g["seed"]), int(g["iters"]), bool(g["tied"])
    Prior = TiedNormalGammas if tied else StackedNormalGammas
    Comp = TiedGaussiansWithNormalGammas if tied else StackedGaussiansWithNormalGammas
    Lik = TiedGaussiansWithDiagonalPrecision if tied else StackedGaussiansWithDiagonalPrecision
    ng = lambda pre: tuple(g[f"{pre}_{k}"] for k in ("mus", "kappas", "alphas", "betas"))
    kind, gating = make_gating(g, K)
    prior = Prior(size=K, dim=D, **{k: g["prior_" + k] for k in ("mus", "kappas", "alphas", "betas")})
    npr.seed(seed + 1)
    comps = Comp(size=K, dim=D, prior=prior, engine=engine)
    m = BayesianMixtureOfGaussians(gating=gating, components=comps, engine=engine)
    assert rel_err(comps.likelihood.mus, g["init_mus"]) < 1e-12
    assert rel_err(comps.likelihood.lmbdas_diags, g["init_lmbdas_diags"]) < 1e-12
    assert rel_err(comps.likelihood.log_likelihood(X), g["loglik_table"]) < tol
    npr.seed(seed + 2)
    m.resample(X, init_labels='random', maxiter=3, progress_bar=False, label_rng='host')
    assert rel_err(m.components.likelihood.mus, g["gibbs_mus"]) < tol
    assert rel_err(m.components.likelihood.lmbdas_diags, g["gibbs_lmbdas_diags"]) < tol
    assert rel_err(m.gating.likelihood.probs, g["gibbs_probs"]) < tol
    for a, b in zip(m.components.posterior.params, ng("gibbs_post")):
        assert rel_err(a, b) < tol
    npr.seed(seed + 3)
    vlb = m.meanfield_coordinate_descent(X, randomize=False, maxiter=iters, tol=0., progress_bar=False)
    assert rel_err(np.array(vlb), g["vi_vlb"]) < tol
    for a, b in zip(m.components.posterior.params, ng("vi_post")):
        assert rel_err(a, b) < 1e-6
    assert rel_err(m.components.expected_log_likelihood(X), g["vi_ell_table"]) < 1e-6
    assert rel_err(m.expected_responsibilities(X), g["vi_resp"]) < 1e-6
    st = m.components.likelihood.weighted_statistics(X, g["vi_resp"])
    assert rel_err(st[0], g["vi_stats_x"]) < tol and rel_err(st[1], g["vi_stats_nd"]) < tol
    assert rel_err(st[3], g["vi_stats_xx"]) < tol and st[1].shape == (K, D)
    import random
    npr.seed(seed + 4)
    random.seed(seed + 14)
    vlb = m.meanfield_stochastic_descent(X, randomize=False, maxiter=iters, step_size=5e-1, batch_size=64,
                                         progress_bar=False)
    assert rel_err(np.array(vlb), g["svi_vlb"]) < tol
    for a, b in zip(m.components.posterior.params, ng("svi_post")):
        assert rel_err(a, b) < 1e-6
    npr.seed(seed + 5)
    lp = m.max_aposteriori(X, randomize=True, maxiter=iters, progress_bar=False)
    assert rel_err(np.array(lp), g["map_logprob"]) < tol
    assert rel_err(m.components.likelihood.mus, g["map_mus"]) < 1e-6
    assert rel_err(m.components.likelihood.lmbdas_diags, g["map_lmbdas_diags"]) < 1e-6
    lik = MixtureOfGaussians(gating=Categorical(dim=K), components=Lik(K, D, engine=engine), engine=engine)
    npr.seed(seed + 6)
    ll = lik.max_likelihood(X, randomize=True, maxiter=iters, progress_bar=False)
    assert rel_err(np.array(ll), g["em_loglik"]) < tol
    assert rel_err(lik.components.mus, g["em_mus"]) < 1e-6
    assert rel_err(lik.components.lmbdas_diags, g["em_lmbdas_diags"]) < 1e-6
    # the conjugate update with the setter defect switched off moves the Gamma factors too
    fixed = Prior(size=K, dim=D, reference_setters=False, **{k: g["prior_" + k] for k in ("mus", "kappas", "alphas", "betas")})
    fixed.nat_param = prior.nat_param + st
    assert np.all(fixed.alphas >= prior.alphas) and np.any(fixed.alphas > prior.alphas + 1.) and np.all(fixed.betas > 0.)


def check_hier_gmm(name, engine, tol=1e-8):
    """Hierarchical mixtures (examples/hgmm): seeded Gibbs sweeps, VI traces without and
    with per-row weights, the natural-gradient driver, tables, and mixtures of mixtures (VI, SVI, Gibbs, EM)."""
    import random
    from mimo_amd.distributions import (NormalWishart, TiedGaussiansWithScaledPrecision,
                                        TiedGaussiansWithHierarchicalNormalWisharts, Categorical)
    from mimo_amd.mixtures import (BayesianMixtureOfGaussiansWithHierarchicalPrior, MixtureOfMixtureOfGaussians,
                                   BayesianMixtureOfMixtureOfGaussians, MixtureOfGaussians)
    g = load_golden(name)
    X, w = g["X"], g["w"]
    K, M, D, seed, iters, sub = (int(g[k]) for k in ("K", "M", "D", "seed", "iters", "sub"))

    def inner(k):
        gating = CategoricalWithDirichlet(dim=k, prior=Dirichlet(dim=k, alphas=np.ones((k,))))
        hyper = NormalWishart(dim=D, mu=np.zeros((D,)), kappa=1e-2, psi=np.eye(D), nu=D + 1. + 1e-8)
        prior = TiedGaussiansWithScaledPrecision(size=k, dim=D, kappas=1e-2 * np.ones((k,)))
        comps = TiedGaussiansWithHierarchicalNormalWisharts(size=k, dim=D, hyper_prior=hyper, prior=prior, engine=engine)
        return BayesianMixtureOfGaussiansWithHierarchicalPrior(size=k, dim=D, gating=gating, components=comps, engine=engine)

    def check_state(m, pre, t):
        c = m.components
        assert rel_err(c.posterior.mus, g[pre + "_post_mus"]) < t, pre
        assert rel_err(c.posterior.kappas, g[pre + "_post_kappas"]) < t, pre
        assert rel_err(c.posterior.lmbdas, g[pre + "_post_lmbdas"]) < t, pre
        for nm, v in zip(("mu", "kappa", "psi", "nu"), c.hyper_posterior.params):
            assert rel_err(np.asarray(v), g[pre + "_hyper_" + nm]) < t, (pre, nm)
        assert rel_err(c.likelihood.mus, g[pre + "_lik_mus"]) < t and rel_err(c.likelihood.lmbdas, g[pre + "_lik_lmbdas"]) < t
        assert rel_err(m.gating.posterior.alphas, g[pre + "_galphas"]) < t, pre

    npr.seed(seed + 1)
    m = inner(K)
    assert rel_err(m.components.likelihood.mus, g["init_lik_mus"]) < 1e-12
    assert rel_err(m.components.likelihood.lmbdas, g["init_lik_lmbdas"]) < 1e-12
    npr.seed(seed + 2)
    m.resample(X, maxiter=iters, maxsubiter=sub, progress_bar=False)
    check_state(m, "gibbs", tol)
    npr.seed(seed + 3)
    vlb = m.meanfield_coordinate_descent(X, randomize=False, maxiter=iters, maxsubiter=sub, tol=0., progress_bar=False)
    assert rel_err(np.array(vlb), g["vi_vlb"]) < tol
    check_state(m, "vi", 1e-7)
    assert rel_err(m.components.expected_log_likelihood(X), g["vi_ell_table"]) < 1e-7
    assert rel_err(m.expected_responsibilities(X), g["vi_resp"]) < 1e-7
    assert rel_err(np.sum(m.components.variational_lowerbound()), np.sum(g["vi_comp_vlb"])) < 1e-7
    assert rel_err(m.components.log_posterior_predictive_gaussian(X), g["vi_logpred"]) < 1e-7
    # explicit-responsibility bound = the fused one (hgmm.py:301-306)
    assert abs(m.variational_lowerbound(X, g["vi_resp"]) - g["vi_vlb"][-1]) < 1e-7 * abs(g["vi_vlb"][-1])
    npr.seed(seed + 4)
    vlb = m.meanfield_coordinate_descent(X, randomize=True, weights=w, maxiter=iters, maxsubiter=sub, tol=0.,
                                         progress_bar=False)
    assert rel_err(np.array(vlb), g["viw_vlb"]) < tol
    check_state(m, "viw", 1e-7)
    npr.seed(seed + 5)
    assert m.meanfield_stochastic_descent(X, randomize=False, weights=w, maxiter=iters, maxsubiter=sub, step_size=5e-1,
                                          progress_bar=False) == []
    check_state(m, "svi", 1e-7)

    def outer():
        gating = CategoricalWithDirichlet(dim=M, prior=Dirichlet(dim=M, alphas=np.ones((M,))))
        return BayesianMixtureOfMixtureOfGaussians(cluster_size=M, mixture_size=K, dim=D, gating=gating,
                                                   components=[inner(K) for _ in range(M)])

    def check_mom(mm, pre, t):
        assert rel_err(mm.gating.posterior.alphas, g[pre + "_galphas"]) < t, pre
        assert rel_err(np.stack([c.components.posterior.mus for c in mm.components]), g[pre + "_post_mus"]) < t, pre
        assert rel_err(np.stack([c.components.hyper_posterior.wishart.psi for c in mm.components]),
                       g[pre + "_hyper_psi"]) < t, pre
        assert rel_err(np.stack([c.gating.posterior.alphas for c in mm.components]), g[pre + "_inner_galphas"]) < t, pre

    npr.seed(seed + 6)
    mm = outer()
    npr.seed(seed + 7)
    mm.meanfield_coordinate_descent(X, randomize=True, maxiter=3, maxsubiter=2, maxsubsubiter=2, progress_bar=False)
    check_mom(mm, "mom_vi", 1e-7)
    assert rel_err(mm.expected_responsibilities(X), g["mom_vi_resp"]) < 1e-7
    npr.seed(seed + 8)
    random.seed(seed + 18)
    mm.meanfield_stochastic_descent(X, randomize=False, maxiter=3, maxsubiter=2, maxsubsubiter=2, step_size=5e-1,
                                    batch_size=64, progress_bar=False)
    check_mom(mm, "mom_svi", 1e-7)
    npr.seed(seed + 9)
    mm.resample(X, init_labels='random', maxiter=2, maxsubiter=2, maxsubsubiter=2, progress_bar=False)
    check_mom(mm, "mom_gibbs", 1e-7)
    assert rel_err(np.stack([c.components.likelihood.mus for c in mm.components]), g["mom_gibbs_lik_mus"]) < 1e-7
    npr.seed(seed + 10)
    comps = [MixtureOfGaussians(gating=Categorical(dim=K), components=TiedGaussiansWithPrecision(K, D, engine=engine),
                                engine=engine) for _ in range(M)]
    em = MixtureOfMixtureOfGaussians(cluster_size=M, mixture_size=K, dim=D, gating=Categorical(dim=M), components=comps)
    ll = em.max_likelihood(X, randomize=True, maxiter=3, maxsubiter=2, progress_bar=False)
    assert rel_err(np.array(ll), g["mom_em_loglik"]) < tol
    assert rel_err(np.stack([c.components.mus for c in em.components]), g["mom_em_mus"]) < 1e-7
    assert rel_err(em.gating.probs, g["mom_em_probs"]) < 1e-7


def check_hier_ilr(name, engine, tol=1e-7):
    """Tied-activation mixture of linear-Gaussian experts (hilr.py:79-290, bayesian.py:1222-1522): seeded Gibbs sweeps,
    VI without and with per-row weights, tables, bounds; the stochastic driver raises like the reference."""
    from mimo_amd.distributions import (NormalWishart, TiedGaussiansWithScaledPrecision, Wishart,
                                        TiedGaussiansWithHierarchicalNormalWisharts, MatrixNormalWithPrecision,
                                        TiedAffineLinearGaussiansWithMatrixNormalWisharts)
    from mimo_amd.mixtures import BayesianMixtureOfLinearGaussiansWithTiedActivation
    g = load_golden(name)
    X, Y, w = g["X"], g["Y"], g["w"]
    K, dx, dy, seed, iters, sub = (int(g[k]) for k in ("K", "dx", "dy", "seed", "iters", "sub"))

    def build():
        gating = CategoricalWithDirichlet(dim=K, prior=Dirichlet(dim=K, alphas=np.ones((K,))))
        bh = NormalWishart(dim=dx, mu=np.zeros((dx,)), kappa=1e-2, psi=np.eye(dx), nu=dx + 1. + 1e-8)
        bp = TiedGaussiansWithScaledPrecision(size=K, dim=dx, kappas=1e-2 * np.ones((K,)))
        basis = TiedGaussiansWithHierarchicalNormalWisharts(size=K, dim=dx, hyper_prior=bh, prior=bp, engine=engine)
        sp = MatrixNormalWithPrecision(column_dim=dx, row_dim=dy, M=np.zeros((dy, dx)), K=1e-2 * np.eye(dx))
        op = TiedGaussiansWithScaledPrecision(size=K, dim=dy, mus=np.zeros((K, dy)), kappas=1e-2 * np.ones((K,)))
        pp = Wishart(dim=dy, psi=np.eye(dy), nu=dy + 1. + 1e-16)
        models = TiedAffineLinearGaussiansWithMatrixNormalWisharts(size=K, column_dim=dx, row_dim=dy, slope_prior=sp,
                                                                   offset_prior=op, precision_prior=pp, engine=engine)
        return BayesianMixtureOfLinearGaussiansWithTiedActivation(size=K, input_dim=dx, output_dim=dy, gating=gating,
                                                                  basis=basis, models=models, engine=engine)

    def check_state(m, pre, t):
        mo, ba = m.models, m.basis
        pairs = [(mo.slope_posterior.M, "_slope_M"), (mo.slope_posterior.K, "_slope_K"), (mo.precision_posterior.psi, "_prec_psi"),
                 (np.asarray(mo.precision_posterior.nu), "_prec_nu"), (mo.offset_posterior.mus, "_off_mus"),
                 (mo.offset_posterior.kappas, "_off_kappas"), (mo.offset_posterior.lmbdas, "_off_lmbdas"),
                 (mo.likelihood.As, "_lik_As"), (mo.likelihood.cs, "_lik_cs"), (mo.likelihood.lmbdas, "_lik_lmbdas"),
                 (ba.posterior.mus, "_basis_mus"), (ba.hyper_posterior.wishart.psi, "_basis_hyper_psi"),
                 (m.gating.posterior.alphas, "_galphas")]
        for a, key in pairs:
            assert rel_err(a, g[pre + key]) < t, (pre, key, rel_err(a, g[pre + key]))

    npr.seed(seed + 1)
    m = build()
    check_state(m, "init", 1e-12)
    assert rel_err(m.models.likelihood.log_likelihood(X, Y), g["init_loglik"]) < 1e-9
    npr.seed(seed + 2)
    m.resample(X, Y, maxiter=iters, maxsubiter=sub, progress_bar=False)
    check_state(m, "gibbs", tol)
    npr.seed(seed + 3)
    assert m.meanfield_coordinate_descent(X, Y, randomize=False, maxiter=iters, maxsubiter=sub, progress_bar=False) == []
    assert g["vi_return"].size == 0
    check_state(m, "vi", tol)
    assert rel_err(m.models.expected_log_likelihood(X, Y), g["vi_models_ell"]) < tol
    assert rel_err(m.basis.expected_log_likelihood(X), g["vi_basis_ell"]) < tol
    assert rel_err(m.expected_responsibilities(X, Y), g["vi_resp"]) < tol
    assert rel_err(np.asarray(m.models.variational_lowerbound()), g["vi_models_vlb"]) < tol
    assert abs(m.variational_lowerbound(X, Y, g["vi_resp"]) - float(g["vi_vlb"])) < tol * abs(float(g["vi_vlb"]))
    npr.seed(seed + 4)
    m.meanfield_coordinate_descent(X, Y, randomize=True, weights=w, maxiter=iters, maxsubiter=sub, progress_bar=False)
    check_state(m, "viw", tol)
    # the bound that comes with the fused pass equals the explicit one (same posterior, unweighted responsibilities)
    bound = m.meanfield_coordinate_descent(X, Y, randomize=False, maxiter=2, maxsubiter=sub, progress_bar=False,
                                           record_bound=True)
    explicit = m.variational_lowerbound(X, Y, m.expected_responsibilities(X, Y))
    assert len(bound) == 2 and abs(bound[-1] - explicit) < 1e-9 * abs(explicit)
    assert bool(g["svi_raises"])
    with pytest.raises(NotImplementedError):
        m.meanfield_stochastic_descent(X, Y, randomize=False, maxiter=2, maxsubiter=2, progress_bar=False)

    # mixture of M such mixtures (hilr.py:293-609): scaled data, VI, prediction through ONE mimo_predict, Gibbs
    from mimo_amd.mixtures import BayesianMixtureOfMixtureOfLinearGaussians
    M = int(g["M"])

    def check_mom(mm, pre, t):
        pairs = [(mm.gating.posterior.alphas, "_galphas"),
                 (np.stack([c.models.slope_posterior.M for c in mm.components]), "_slope_M"),
                 (np.stack([c.models.offset_posterior.mus for c in mm.components]), "_off_mus"),
                 (np.stack([c.models.precision_posterior.psi for c in mm.components]), "_prec_psi"),
                 (np.stack([c.basis.posterior.mus for c in mm.components]), "_basis_mus"),
                 (np.stack([c.gating.posterior.alphas for c in mm.components]), "_inner_galphas")]
        for a, key in pairs:
            assert rel_err(a, g[pre + key]) < t, (pre, key, rel_err(a, g[pre + key]))

    npr.seed(seed + 6)
    gating = CategoricalWithDirichlet(dim=M, prior=Dirichlet(dim=M, alphas=np.ones((M,))))
    mm = BayesianMixtureOfMixtureOfLinearGaussians(cluster_size=M, mixture_size=K, input_dim=dx, output_dim=dy, gating=gating,
                                                   components=[build() for _ in range(M)])
    mm.init_transform(X, Y)
    npr.seed(seed + 7)
    assert mm.meanfield_coordinate_descent(X, Y, randomize=True, maxiter=3, maxsubiter=2, maxsubsubiter=2,
                                           progress_bar=False) == []
    check_mom(mm, "mom_vi", 1e-6)
    xx = mm.input_transform.transform(X)
    assert rel_err(mm.meanfield_predictive_weights(xx), g["mom_pred_weights"]) < 1e-6
    assert rel_err(mm.meanfield_predictive_activation(X), g["mom_pred_activation"]) < 1e-6
    mus, covars = mm.meanfield_predictive_moments(xx)
    assert rel_err(mus, g["mom_pred_mus"]) < 1e-6 and rel_err(covars, g["mom_pred_covars"]) < 1e-6
    for pred in ("average", "mode"):
        mu, var, std = mm.meanfield_prediction(X, prediction=pred)
        assert rel_err(mu, g[f"mom_pred_{pred}_mu"]) < 1e-6, pred
        assert rel_err(var, g[f"mom_pred_{pred}_var"]) < 1e-6 and rel_err(std, g[f"mom_pred_{pred}_std"]) < 1e-6, pred
    assert rel_err(mm.meanfield_prediction(X, prediction='average', variance='full')[1], g["mom_pred_average_covar"]) < 1e-6
    npr.seed(seed + 8)
    mm.resample(X, Y, init_labels='random', maxiter=2, maxsubiter=2, maxsubsubiter=2, progress_bar=False)
    check_mom(mm, "mom_gibbs", 1e-6)
    assert bool(g["mom_svi_raises"])
    with pytest.raises(NotImplementedError):
        mm.meanfield_stochastic_descent(X, Y, randomize=False, maxiter=1, maxsubiter=1, maxsubsubiter=1, batch_size=32,
                                        progress_bar=False)
    with pytest.raises(NotImplementedError):
        mm.likelihood.max_likelihood(X, Y)


def check_tied_ilr_prediction(name, engine, tol=1e-7):
    """examples/ilr/evaluate_sine.py at fixture size: tied MNW experts, Gibbs -> SVI -> VI -> prediction."""
    import random
    g = load_golden(name)
    K, seed = int(g["K"]), int(g["seed"])
    X, Y, Xtr, Ytr = g["X"], g["Y"], g["Xtr"], g["Ytr"]
    dx, dy = X.shape[1], Y.shape[1]
    kind, gating = make_gating(g, K)
    bprior = StackedNormalWisharts(size=K, dim=dx, **{k: g["bprior_" + k] for k in ("mus", "kappas", "psis", "nus")})
    npr.seed(seed + 1)
    basis = StackedGaussiansWithNormalWisharts(size=K, dim=dx, prior=bprior, engine=engine)
    mprior = TiedMatrixNormalWisharts(K, dx + 1, dy, **{k: g["mprior_" + k] for k in ("Ms", "Ks", "psis", "nus")})
    models = TiedLinearGaussiansWithMatrixNormalWisharts(K, dx + 1, dy, mprior, affine=True, engine=engine)
    ilr = BayesianMixtureOfLinearGaussians(size=K, input_dim=dx, output_dim=dy, gating=gating, basis=basis,
                                           models=models, engine=engine)
    ilr.init_transform(Xtr, Ytr)
    npr.seed(seed + 2)
    random.seed(seed + 3)
    ilr.resample(Xtr, Ytr, init_labels='random', maxiter=int(g["gibbs_iters"]), progress_bar=False, label_rng='host')
    assert rel_err(ilr.models.likelihood.As, g["gibbs_As"]) < tol
    assert rel_err(ilr.models.likelihood.lmbdas, g["gibbs_lmbdas"]) < tol
    for a, b in zip(ilr.models.posterior.params, mnw_of(g, "gibbs_mpost")):
        assert rel_err(a, b) < tol
    vlb = ilr.meanfield_stochastic_descent(Xtr, Ytr, randomize=False, maxiter=int(g["svi_iters"]), step_size=5e-1,
                                           batch_size=64, progress_bar=False)
    assert rel_err(np.array(vlb), g["svi_vlb"]) < tol
    for a, b in zip(ilr.models.posterior.params, mnw_of(g, "svi_mpost")):
        assert rel_err(a, b) < 1e-6
    vlb = ilr.meanfield_coordinate_descent(Xtr, Ytr, randomize=False, maxiter=int(g["vi_iters"]), tol=0.,
                                           progress_bar=False)
    assert rel_err(np.array(vlb), g["vi_vlb"]) < tol
    for a, b in zip(ilr.models.posterior.params, mnw_of(g, "vi_mpost")):
        assert rel_err(a, b) < 1e-6
    for a, b in zip(ilr.basis.posterior.params, nw_of(g, "vi_bpost")):
        assert rel_err(a, b) < 1e-6
    check_prediction(ilr, g, tol=1e-6)


def check_prediction(ilr, g, tol):
    """meanfield_prediction and its helpers (ilr.py:325-430) against the reference's outputs."""
    X, Y = g["X"], g["Y"]
    xx = ilr.input_transform.transform(X)
    assert rel_err(ilr.basis.log_posterior_predictive_gaussian(xx), g["basis_logpred_gaussian"]) < tol
    assert rel_err(ilr.meanfield_predictive_weights(xx), g["pred_weights_gaussian"]) < tol
    assert rel_err(ilr.meanfield_predictive_activation(X), g["pred_activation_gaussian"]) < tol
    mus, covars = ilr.meanfield_predictive_moments(xx)
    assert rel_err(mus, g["pred_mus_gaussian"]) < tol and rel_err(covars, g["pred_covars_gaussian"]) < tol
    for pred in ("average", "mode"):
        mu, var, std = ilr.meanfield_prediction(X, prediction=pred)
        assert rel_err(mu, g[f"pred_{pred}_gaussian_mu"]) < tol, pred
        assert rel_err(var, g[f"pred_{pred}_gaussian_var"]) < tol, pred
        assert rel_err(std, g[f"pred_{pred}_gaussian_std"]) < tol, pred
    mu, covar, std = ilr.meanfield_prediction(X, prediction='average', variance='full')
    assert rel_err(covar, g["pred_average_gaussian_covar"]) < tol
    # nlpd: no reference vector (the reference raises); consistent with its own tables
    from scipy.special import logsumexp
    mu, var, std, nlpd = ilr.meanfield_prediction(X, Y, prediction='average')
    yy = ilr.output_transform.transform(Y)
    log_pl = ilr.meanfiled_log_predictive_likelihood(xx, yy)
    ref = - logsumexp(log_pl + np.log(ilr.meanfield_predictive_weights(xx) + np.finfo(float).tiny), axis=0)
    assert rel_err(nlpd, ref) < tol
    try:
        ilr.meanfield_prediction(X, dist='studentt')
    except NotImplementedError:
        pass
    else:
        raise AssertionError("studentt must be rejected")


def check_nan_rows(name, engine, tol=1e-9):
    """Rows that hold a NaN, through the reference-shaped methods, against the reference's outputs: normaliser-only
    log-density (tables, responsibilities, labels carry the row), no contribution to the component statistics, full
    contribution to the gating counts (S.gating_counts)."""
    from mimo_amd.distributions import (Dirichlet, CategoricalWithDirichlet, StackedNormalWisharts,
                                        StackedGaussiansWithNormalWisharts)
    from mimo_amd.mixtures import BayesianMixtureOfGaussians
    g = load_golden(name)
    X, K, D = g["X"], int(g["K"]), int(g["D"])
    gating = CategoricalWithDirichlet(dim=K, prior=Dirichlet(dim=K, alphas=np.ones(K)))
    prior = StackedNormalWisharts(size=K, dim=D, mus=np.zeros((K, D)), kappas=1e-2 * np.ones(K),
                                  psis=np.stack(K * [np.eye(D)]), nus=(D + 1.) * np.ones(K) + 1e-8)
    comps = StackedGaussiansWithNormalWisharts(size=K, dim=D, prior=prior, engine=engine)
    model = BayesianMixtureOfGaussians(gating=gating, components=comps, engine=engine)
    model.components.likelihood.params = (g["lik_mus"], g["lik_lmbdas"])
    model.gating.likelihood.params = g["lik_probs"].copy()
    lik = model.components.likelihood
    assert rel_err(lik.log_likelihood(X.copy()), g["A1_loglik"]) < tol
    assert rel_err(model.likelihood.log_complete_likelihood(X.copy()), g["A2_lcl"]) < tol
    assert rel_err(model.likelihood.responsibilities(X.copy()), g["A2_resp"]) < tol
    assert rel_err(model.likelihood.log_likelihood(X.copy()), g["A2_ll"]) < tol
    for resp, pre in ((g["A2_resp"], "stats"), (g["resp0"], "stats0")):
        st = lik.weighted_statistics(X.copy(), resp)
        assert rel_err(st[0], g[pre + "_xk"]) < tol and rel_err(st[1], g[pre + "_nk"]) < tol and rel_err(st[2], g[pre + "_xxTk"]) < tol
    # the fused passes: statistics of the E-step's own responsibilities / labels, gating counts over all rows
    eng = model.likelihood._bind(X.copy())
    assert eng.n_bad == len(g["bad"]) and np.array_equal(eng.nan_rows(), g["bad"])
    S, sc = eng.estep(*model.likelihood.canonical())
    assert rel_err(S.sx, g["stats_xk"]) < tol and rel_err(S.n, g["stats_nk"]) < tol and rel_err(S.sxx, g["stats_xxTk"]) < tol
    assert rel_err(S.gating_counts, g["counts"]) < tol and abs(sc[0] - g["A2_ll"].sum()) < tol * abs(g["A2_ll"].sum())
    labels, Sl = eng.gibbs_labels(*model.likelihood.canonical(), u=g["u"])
    assert np.array_equal(labels, g["labels"])
    assert rel_err(Sl.sx, g["lstats_xk"]) < tol and np.array_equal(Sl.n, g["lstats_nk"]) and rel_err(Sl.sxx, g["lstats_xxTk"]) < tol
    assert np.array_equal(Sl.gating_counts, g["lcounts"])
    Sl2 = eng.label_stats(g["labels"], K)
    assert np.array_equal(Sl2.n, g["lstats_nk"]) and rel_err(Sl2.sxx, g["lstats_xxTk"]) < tol and np.array_equal(Sl2.gating_counts, g["lcounts"])
    # a driver runs end to end on such data (the reference's own drivers do not: its mean-field table drops the rows,
    # bayesian.py:296 with gaussian.py:468-469, and the statistics call then fails on the shapes)
    vlb = model.meanfield_coordinate_descent(X.copy(), randomize=False, maxiter=5, tol=0., progress_bar=False)
    assert np.all(np.isfinite(vlb)) and np.all(np.diff(vlb) > -1e-8 * abs(vlb[-1]))


def check_nan_rows_ilr(name, engine, tol=1e-9):
    """Rows with a NaN in x, in y, or in both through the linear-Gaussian mixture's reference-shaped methods, against the reference's
    outputs (lingauss.py:103-104, 150-151, 306-310, 330-345; ilr.py:71-84, 161-164): element-wise nan_to_num in the experts'
    density, its data part zeroed only where x AND y hold a NaN — and nowhere inside log_complete_likelihood, where the input
    density has nan_to_num'ed x in place by then —, every such row dropped from the statistics."""
    from mimo_amd.distributions import (Dirichlet, CategoricalWithDirichlet, StackedNormalWisharts, StackedGaussiansWithNormalWisharts,
                                        StackedMatrixNormalWisharts, StackedLinearGaussiansWithMatrixNormalWisharts)
    from mimo_amd.mixtures import BayesianMixtureOfLinearGaussians
    g = load_golden(name)
    X, Y, K = g["X"], g["Y"], int(g["K"])
    dx, dy = X.shape[1], Y.shape[1]
    gating = CategoricalWithDirichlet(dim=K, prior=Dirichlet(dim=K, alphas=np.ones(K)))
    bprior = StackedNormalWisharts(size=K, dim=dx, mus=np.zeros((K, dx)), kappas=1e-2 * np.ones(K),
                                   psis=np.stack(K * [1e2 * np.eye(dx)]), nus=(dx + 1.) * np.ones(K) + 1e-16)
    basis = StackedGaussiansWithNormalWisharts(size=K, dim=dx, prior=bprior, engine=engine)
    mprior = StackedMatrixNormalWisharts(K, dx + 1, dy, Ms=np.zeros((K, dy, dx + 1)), Ks=np.stack(K * [1e-2 * np.eye(dx + 1)]),
                                         psis=np.stack(K * [np.eye(dy)]), nus=(dy + 1.) * np.ones(K) + 1e-16)
    models = StackedLinearGaussiansWithMatrixNormalWisharts(K, dx + 1, dy, mprior, affine=True, engine=engine)
    ilr = BayesianMixtureOfLinearGaussians(size=K, input_dim=dx, output_dim=dy, gating=gating, basis=basis, models=models,
                                           engine=engine)
    ilr.basis.likelihood.params = (g["lik_mus"], g["lik_lmbdas"])
    ilr.models.likelihood.params = (g["lik_As"], g["lik_lmbdas_y"])
    ilr.gating.likelihood.params = g["lik_probs"].copy()
    bx, by = np.isnan(X).any(axis=1), np.isnan(Y).any(axis=1)
    assert (bx & ~by).any() and (~bx & by).any() and (bx & by).any() and np.array_equal(np.flatnonzero(bx | by), g["bad"])
    assert rel_err(ilr.basis.likelihood.log_likelihood(X.copy()), g["A1_basis_loglik"]) < tol
    assert rel_err(ilr.models.likelihood.log_likelihood(X.copy(), Y.copy()), g["A5_loglik"]) < tol
    assert rel_err(ilr.likelihood.log_complete_likelihood(X.copy(), Y.copy()), g["A7_lcl"]) < tol
    assert rel_err(ilr.likelihood.responsibilities(X.copy(), Y.copy()), g["A7_resp"]) < tol
    assert rel_err(ilr.likelihood.log_likelihood(X.copy(), Y.copy()), g["A7_ll"]) < tol
    ms = ilr.models.likelihood.weighted_statistics(X.copy(), Y.copy(), g["resp0"])
    for a, b in zip(ms, (g["mstats0_yxTk"], g["mstats0_xxTk"], g["mstats0_yyTk"], g["mstats0_nk"])):
        assert rel_err(a, b) < tol
    bs = ilr.basis.likelihood.weighted_statistics(X.copy(), g["resp0"])
    assert rel_err(bs[0], g["bstats0_xk"]) < tol and rel_err(bs[1], g["bstats0_nk"]) < tol and rel_err(bs[2], g["bstats0_xxTk"]) < tol
    import mimo_amd.mixtures.ilr as ilr_mod
    u = g["u"]
    orig = ilr_mod.npr.random
    ilr_mod.npr.random = lambda size=None: u.reshape(size)           # the uniforms the reference drew
    try:
        for lazy in (True, False):
            table, labels = ilr.resample_labels(X.copy(), Y.copy(), lazy=lazy)
            assert np.array_equal(labels, g["labels"])
            assert rel_err(np.asarray(table), g["A7_lcl"]) < tol
    finally:
        ilr_mod.npr.random = orig
