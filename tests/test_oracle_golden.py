"""Pins oracle/mimo_oracle.py (the CPU restatement) against vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only.  Tolerance 1e-12 relative (same arithmetic, same order)."""
import numpy as np
import numpy.random as npr
import pytest

from conftest import (load_golden, rel_err, GMM_CASES, ILR_CASES, GIBBS_CASES, GMM_FULLK_CASES, ILR_FULLK_CASES, gating_of, nw_of,
                      mnw_of)
from oracle import mimo_oracle as O

TOL = 1e-12


@pytest.mark.parametrize("name", GMM_CASES + GMM_FULLK_CASES)
def test_gmm_tables_and_stats(name):
    g = load_golden(name)
    X, K = g["X"], int(g["K"])
    kind, gpost = gating_of(g, "gpost")
    post = nw_of(g, "post")
    # A10 / A12 / A13 from the random responsibilities
    st0 = O.gauss_weighted_statistics(X, g["resp0"])
    assert rel_err(st0[0], g["stats0_xk"]) < TOL and rel_err(st0[2], g["stats0_xxTk"]) < TOL
    assert rel_err(st0[1], g["stats0_nk"]) < TOL
    assert rel_err(O.categorical_weighted_statistics(g["resp0"]), g["counts0"]) < TOL
    post_o = O.stacked_nw_update(nw_of(g, "prior"), st0)
    for a, b in zip(post_o, post):
        assert rel_err(a, b) < 1e-10
    _, gprior = gating_of(g, "gprior")
    gpost_o = O.gating_update(kind, gprior, g["counts0"])
    for a, b in zip(np.atleast_2d(gpost_o), np.atleast_2d(gpost)):
        assert rel_err(a, b) < TOL
    # A1 / A2 (Gibbs / EM form)
    mus, lmbdas, probs = g["lik_mus"], g["lik_lmbdas"], g["lik_probs"]
    assert rel_err(O.gauss_log_likelihood(X, mus, lmbdas), g["A1_loglik"]) < TOL
    lcl = O.gmm_log_complete_likelihood(X, mus, lmbdas, probs)
    assert rel_err(lcl, g["A2_lcl"]) < TOL
    assert rel_err(O.responsibilities(lcl), g["A2_resp"]) < 1e-11
    # A3 / A4 (mean-field form)
    es = [np.stack(v) for v in zip(*[O.nw_expected_statistics(*(p[k] for p in post)) for k in range(K)])]
    for a, b in zip(es, (g["estats_a"], g["estats_b"], g["estats_c"], g["estats_d"])):
        assert rel_err(a, b) < TOL
    assert rel_err(O.gauss_nw_expected_log_likelihood(X, post, chunk=100), g["A3_eloglik"]) < TOL
    elcl = O.gmm_expected_log_complete_likelihood(X, post, kind, gpost, chunk=64)
    assert rel_err(elcl, g["A4_elcl"]) < TOL
    eresp = O.responsibilities(elcl)
    assert rel_err(eresp, g["A4_eresp"]) < 1e-11
    # statistics of those responsibilities, ELBO pieces
    st = O.gauss_weighted_statistics(X, g["A4_eresp"])
    assert rel_err(st[0], g["stats_xk"]) < TOL and rel_err(st[2], g["stats_xxTk"]) < TOL
    assert abs(np.sum(g["A4_eresp"] * g["A3_eloglik"]) - g["vlb_obs"]) < TOL * abs(g["vlb_obs"])
    assert abs(O.vlb_labels(g["A4_eresp"], kind, gpost) - g["vlb_labels"]) < 1e-11 * abs(g["vlb_labels"])
    assert abs(O.gating_vlb(kind, gpost, gprior) - g["vlb_gating"]) < 1e-10 * max(1, abs(g["vlb_gating"]))
    prior = nw_of(g, "prior")
    vc = np.array([O.nw_vlb(tuple(p[k] for p in post), tuple(p[k] for p in prior)) for k in range(K)])
    assert rel_err(vc, g["vlb_comps"]) < 1e-10
    # identity used by the fused kernel: obs + labels terms == sum_n logsumexp_k
    from scipy.special import logsumexp
    ident = np.sum(logsumexp(g["A4_elcl"], axis=0))
    assert abs(ident - (g["vlb_obs"] + g["vlb_labels"])) < 1e-10 * abs(ident)


@pytest.mark.parametrize("name", GMM_CASES + GMM_FULLK_CASES)
def test_gmm_labels_and_update(name):
    g = load_golden(name)
    X, K = g["X"], int(g["K"])
    kind, gpost = gating_of(g, "gpost")
    lcl = O.gmm_log_complete_likelihood(X, g["lik_mus"], g["lik_lmbdas"], g["lik_probs"])
    assert np.array_equal(O.sample_discrete_from_log(lcl, g["u_mt"]), g["labels_mt"])
    u_ph = O.philox_uniforms(1337, np.arange(X.shape[0]), 3)
    assert np.array_equal(O.sample_discrete_from_log(lcl, u_ph), g["labels_philox"])
    oh = O.one_hot(g["labels_mt"], K)
    stl = O.gauss_weighted_statistics(X, oh)
    assert rel_err(stl[2], g["lstats_xxTk"]) < TOL and rel_err(stl[1], g["lstats_nk"]) < TOL
    assert np.array_equal(O.categorical_statistics(g["labels_mt"], K), g["lcounts"])
    post2 = O.stacked_nw_update(nw_of(g, "prior"), O.gauss_weighted_statistics(X, g["A4_eresp"]))
    for a, b in zip(post2, nw_of(g, "post2")):
        assert rel_err(a, b) < 1e-9


@pytest.mark.parametrize("name", GMM_CASES + GMM_FULLK_CASES)
def test_gmm_vi_trace(name):
    g = load_golden(name)
    X = g["X"]
    kind, gpost = gating_of(g, "gpost")
    _, gprior = gating_of(g, "gprior")
    prior = nw_of(g, "prior")
    resp = g["A4_eresp"]
    vlbs = []
    for _ in range(len(g["vi_vlb"])):
        post, gp, resp, v = O.gmm_vi_iteration(X, prior, gprior, kind, resp)
        vlbs.append(v)
    assert rel_err(np.array(vlbs), g["vi_vlb"]) < 1e-9
    for a, b in zip(post, nw_of(g, "vi_post")):
        assert rel_err(a, b) < 1e-7


@pytest.mark.parametrize("name", ILR_CASES + ILR_FULLK_CASES)
def test_ilr_tables_stats_trace(name):
    g = load_golden(name)
    X, Y, K = g["X"], g["Y"], int(g["K"])
    kind, gpost = gating_of(g, "gpost")
    _, gprior = gating_of(g, "gprior")
    bprior, mprior, bpost, mpost = nw_of(g, "bprior"), mnw_of(g, "mprior"), nw_of(g, "bpost"), mnw_of(g, "mpost")
    ms0 = O.lingauss_weighted_statistics(X, Y, g["resp0"])
    for a, b in zip(ms0, (g["mstats0_yxTk"], g["mstats0_xxTk"], g["mstats0_yyTk"], g["mstats0_nk"])):
        assert rel_err(a, b) < TOL
    for a, b in zip(O.stacked_mnw_update(mprior, ms0), mpost):
        assert rel_err(a, b) < 1e-9
    assert rel_err(O.lingauss_log_likelihood(X, Y, g["lik_As"], g["lik_lmbdas_y"]), g["A5_loglik"]) < TOL
    lcl = O.ilr_log_complete_likelihood(X, Y, g["lik_mus"], g["lik_lmbdas"], g["lik_As"], g["lik_lmbdas_y"],
                                        g["lik_probs"])
    assert rel_err(lcl, g["A7_lcl"]) < TOL
    es = [np.stack(v) for v in zip(*[O.mnw_expected_statistics(*(p[k] for p in mpost)) for k in range(K)])]
    for a, b in zip(es, (g["mestats_a"], g["mestats_b"], g["mestats_c"], g["mestats_d"])):
        assert rel_err(a, b) < 1e-11
    assert rel_err(O.lingauss_mnw_expected_log_likelihood(X, Y, mpost, chunk=100), g["A6_eloglik"]) < 1e-11
    elcl = O.ilr_expected_log_complete_likelihood(X, Y, bpost, mpost, kind, gpost)
    assert rel_err(elcl, g["A7_elcl"]) < 1e-11
    assert rel_err(O.responsibilities(elcl), g["A7_eresp"]) < 1e-10
    ms = O.lingauss_weighted_statistics(X, Y, g["A7_eresp"])
    for a, b in zip(ms, (g["mstats_yxTk"], g["mstats_xxTk"], g["mstats_yyTk"], g["mstats_nk"])):
        assert rel_err(a, b) < TOL
    vm = np.array([O.mnw_vlb(tuple(p[k] for p in mpost), tuple(p[k] for p in mprior)) for k in range(K)])
    assert rel_err(vm, g["vlb_models"]) < 1e-9
    assert np.array_equal(O.sample_discrete_from_log(lcl, g["u_mt"]), g["labels_mt"])
    assert np.array_equal(O.sample_discrete_from_log(lcl, O.philox_uniforms(1337, np.arange(len(X)), 3)),
                          g["labels_philox"])
    resp, vlbs = g["A7_eresp"], []
    for _ in range(len(g["vi_vlb"])):
        bp, mp, gp, resp, v = O.ilr_vi_iteration(X, Y, bprior, mprior, gprior, kind, resp)
        vlbs.append(v)
    assert rel_err(np.array(vlbs), g["vi_vlb"]) < 1e-8
    for a, b in zip(mp, mnw_of(g, "vi_mpost")):
        assert rel_err(a, b) < 1e-6


@pytest.mark.parametrize("name", GIBBS_CASES)
def test_gibbs_trace_rng_order(name):
    """Seeded Gibbs trace: same host-RNG call order as gmm.py:207-225 => identical labels every sweep."""
    g = load_golden(name)
    X, K = g["X"], int(g["K"])
    kind, gprior = gating_of(g, "gprior")
    prior = nw_of(g, "prior")
    npr.seed(int(g["seed2"]))
    labels = npr.choice(a=K, p=g["lik0_probs"], size=len(X))        # categorical.py:32
    assert np.array_equal(labels, g["labels_init"])
    s = 0
    while f"s{s}_labels" in g:
        post, gpost, mus, lmbdas, probs, labels = O.gmm_gibbs_sweep(X, prior, gprior, kind, labels, npr)
        assert rel_err(mus, g[f"s{s}_mus"]) < 1e-9
        assert rel_err(lmbdas, g[f"s{s}_lmbdas"]) < 1e-9
        assert rel_err(probs, g[f"s{s}_probs"]) < 1e-12
        assert np.array_equal(labels, g[f"s{s}_labels"])
        s += 1
    assert s == 5
    assert rel_err(mus, g["driver_mus"]) < 1e-9


def test_philox_known_answer():
    """Philox4x32-10 known-answer vectors of Random123 (kat_vectors): counter/key all zero, all ones."""
    def raw(c, k):
        M = 0xFFFFFFFF
        c = list(c); k = list(k)
        for _ in range(10):
            p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
            c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & M, (p0 >> 32) ^ c[3] ^ k[1], p0 & M]
            k = [(k[0] + 0x9E3779B9) & M, (k[1] + 0xBB67AE85) & M]
        return c
    assert raw([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert raw([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    # the vectorised oracle agrees with the scalar rounds
    c = raw([5, 0, 3, 0], [1337, 0])
    u = ((c[0] >> 5) * 67108864.0 + (c[1] >> 6)) / 9007199254740992.0
    assert O.philox_uniforms(1337, np.array([5]), 3)[0] == u


@pytest.mark.parametrize("name", ["tied_ilr_sine_k8", "tied_ilr_dx3_dy2_k6"])
def test_prediction_restatement(name):
    """ilr.py:339-430 / bayesian.py:303-313,949-962 restated in the oracle vs the reference's outputs
    (posterior taken from the fixture, inputs in model coordinates)."""
    g = load_golden(name)
    X, Xtr, Ytr = g["X"], g["Xtr"], g["Ytr"]
    xx = (X - Xtr.mean(0)) / Xtr.std(0)
    bpost, mpost = nw_of(g, "vi_bpost"), mnw_of(g, "vi_mpost")
    gmean = O.stick_mean(g["vi_gpost_gammas"], g["vi_gpost_deltas"])
    assert rel_err(O.stacked_mvn_logpdf(xx, *O.nw_posterior_predictive_gaussian(bpost)), g["basis_logpred_gaussian"]) < 1e-10
    assert rel_err(O.ilr_predictive_weights(xx, bpost, gmean), g["pred_weights_gaussian"]) < 1e-10
    mus, lmbdas = O.mnw_posterior_predictive_gaussian(xx, mpost)
    assert rel_err(mus, g["pred_mus_gaussian"]) < 1e-10
    assert rel_err(np.linalg.inv(lmbdas), g["pred_covars_gaussian"]) < 1e-10
    sd = Ytr.std(0)
    for pred in ("average", "mode"):
        mu, covar, _ = O.ilr_meanfield_prediction(xx, bpost, mpost, gmean, prediction=pred)
        assert rel_err(mu * sd + Ytr.mean(0), g[f"pred_{pred}_gaussian_mu"]) < 1e-10
        var = np.einsum('ndd->nd', covar) * Ytr.var(0)
        assert rel_err(var, g[f"pred_{pred}_gaussian_var"]) < 1e-9
    # the canonical-level statement of mimo_predict equals the reference-shaped one
    Ms, Ks, psis, nus = mpost
    dy = Ms.shape[1]
    P = (nus - dy + 1)[:, None, None] * psis
    pm, lm = O.nw_posterior_predictive_gaussian(bpost)
    b = np.einsum('kdl,kl->kd', lm, pm)
    c = -0.5 * np.einsum('kd,kd->k', pm, b) - 0.5 * xx.shape[1] * np.log(2 * np.pi) + 0.5 * np.linalg.slogdet(lm)[1] + np.log(gmean)
    yy = (g["Y"] - Ytr.mean(0)) / sd
    for pred in ("average", "mode"):
        a1 = O.ilr_meanfield_prediction(xx, bpost, mpost, gmean, prediction=pred, y=yy)
        a2 = O.predict_canonical(xx, c, b, lm, Ms, np.linalg.inv(Ks), np.linalg.inv(P), True, pred, yy, P,
                                 np.linalg.slogdet(P)[1])
        for u, v in zip(a1, a2):
            assert rel_err(u, v) < 1e-9


def test_tied_posterior_restatement():
    """composite.py:273-283 / 798-808: pooled Wishart block, from the fixture's own Gibbs posterior."""
    g = load_golden("tied_gmm_d3_k5")
    mus, kappas, psis, nus = nw_of(g, "gibbs_post")
    D = mus.shape[1]
    nat = (kappas[:, None] * mus, kappas, g["gibbs_nat2"], nus - D)
    out = O.tied_nw_nat_to_std(*nat, D)
    for a, b in zip(out, (mus, kappas, psis, nus)):
        assert rel_err(a, b) < 1e-10
    g = load_golden("tied_ilr_sine_k8")
    Ms, Ks, psis, nus = mnw_of(g, "vi_mpost")
    dy, dc = Ms.shape[1], Ms.shape[2]
    nat = (Ms @ Ks, Ks, np.linalg.inv(psis) + Ms @ Ks @ np.swapaxes(Ms, 1, 2), nus - dy - 1 + dc)
    out = O.tied_mnw_nat_to_std(*nat, dy, dc)
    for a, b in zip(out, (Ms, Ks, psis, nus)):
        assert rel_err(a, b) < 1e-9


@pytest.mark.parametrize("name,tied", [("diag_gmm_d3_k5", False), ("tied_diag_gmm_d4_k6", True)])
def test_diagonal_restatement(name, tied):
    """gaussian.py:802-832, bayesian.py:441-455, composite.py:314-382 against the reference's own tables, and the
    reference's observable update rule: the posterior keeps the prior's Gamma factors (composite.py:472-484)."""
    g = load_golden(name)
    X = g["X"]
    ng = lambda pre: tuple(g[f"{pre}_{k}"] for k in ("mus", "kappas", "alphas", "betas"))
    prior, post = ng("prior"), ng("vi_post")
    assert rel_err(O.diag_gauss_log_likelihood(X, g["init_mus"], g["init_lmbdas_diags"]), g["loglik_table"]) < 1e-12
    assert rel_err(O.diag_gauss_ng_expected_log_likelihood(X, post), g["vi_ell_table"]) < 1e-12
    for a, b in zip(O.ng_std_to_nat(*post), g["vi_nat"]):
        assert rel_err(a, b) < 1e-12
    xk, ndk, _, xxk = O.diag_gauss_weighted_statistics(X, g["vi_resp"])
    assert rel_err(xk, g["vi_stats_x"]) < 1e-12 and rel_err(xxk, g["vi_stats_xx"]) < 1e-12
    assert rel_err(ndk, g["vi_stats_nd"]) < 1e-12
    for pre in ("gibbs_post", "vi_post", "svi_post"):
        assert np.array_equal(g[pre + "_alphas"], prior[2]) and np.array_equal(g[pre + "_betas"], prior[3])
    # one more coordinate-ascent step from the fixture's responsibilities reproduces a fixed point of the trace
    upd = O.stacked_ng_update(prior, (xk, ndk, ndk, xxk), (prior[2], prior[3]), tied=tied)
    ell = O.diag_gauss_ng_expected_log_likelihood(X, upd)
    assert np.all(np.isfinite(ell)) and ell.shape == g["vi_ell_table"].shape
    assert np.all(np.isfinite(O.ng_vlb(upd, prior)))


@pytest.mark.parametrize("name", ["nan_rows_gmm_d3_k5", "nan_rows_gmm_d16_k70"])
def test_rows_with_nan_follow_the_reference(name):
    """gaussian.py:493-494, 512-520: rows that hold a NaN keep the normaliser-only log-density and are dropped from the
    statistics; the gating counts keep them.  The oracle's restatement against outputs of the reference."""
    g = load_golden(name)
    X, K = g["X"], int(g["K"])
    L = O.gauss_log_likelihood(X, g["lik_mus"], g["lik_lmbdas"])
    assert rel_err(L, g["A1_loglik"]) < TOL and np.isnan(X).any()
    lcl = O.gmm_log_complete_likelihood(X, g["lik_mus"], g["lik_lmbdas"], g["lik_probs"])
    assert rel_err(lcl, g["A2_lcl"]) < TOL and rel_err(O.responsibilities(lcl), g["A2_resp"]) < TOL
    for resp, pre in ((g["A2_resp"], "stats"), (g["resp0"], "stats0"), (O.one_hot(g["labels"], K), "lstats")):
        xk, nk, xxTk, _ = O.gauss_weighted_statistics(X, resp)
        assert rel_err(xk, g[pre + "_xk"]) < TOL and rel_err(nk, g[pre + "_nk"]) < TOL and rel_err(xxTk, g[pre + "_xxTk"]) < TOL
    assert np.array_equal(O.sample_discrete_from_log(lcl, g["u"]), g["labels"])
    assert rel_err(O.categorical_weighted_statistics(g["A2_resp"]), g["counts"]) < TOL
    assert np.array_equal(O.categorical_statistics(g["labels"], K), g["lcounts"])


def test_rows_with_nan_in_a_linear_gaussian_mixture_follow_the_reference():
    """lingauss.py:103-104, 150-151, 306-310, 330-345 and ilr.py:71-84, 161-164 of the reference on rows with a NaN in x, in y, or in
    both: the oracle's restatement against outputs of the reference (fixture nan_rows_ilr_dx2_dy1_k6)."""
    g = load_golden("nan_rows_ilr_dx2_dy1_k6")
    X, Y, K = g["X"], g["Y"], int(g["K"])
    assert rel_err(O.gauss_log_likelihood(X, g["lik_mus"], g["lik_lmbdas"]), g["A1_basis_loglik"]) < TOL
    assert rel_err(O.lingauss_log_likelihood(X, Y, g["lik_As"], g["lik_lmbdas_y"]), g["A5_loglik"]) < TOL
    lcl = O.ilr_log_complete_likelihood(X, Y, g["lik_mus"], g["lik_lmbdas"], g["lik_As"], g["lik_lmbdas_y"], g["lik_probs"])
    assert rel_err(lcl, g["A7_lcl"]) < TOL and rel_err(O.responsibilities(lcl), g["A7_resp"]) < TOL
    from scipy.special import logsumexp
    assert rel_err(logsumexp(lcl, axis=0), g["A7_ll"]) < TOL
    ms = O.lingauss_weighted_statistics(X, Y, g["resp0"])
    for a, b in zip(ms, (g["mstats0_yxTk"], g["mstats0_xxTk"], g["mstats0_yyTk"], g["mstats0_nk"])):
        assert rel_err(a, b) < TOL
    assert np.array_equal(O.sample_discrete_from_log(lcl, g["u"]), g["labels"])
