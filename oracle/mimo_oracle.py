"""CPU ORACLE — test infrastructure only, never part of the product path.

A NumPy restatement of the hot path of hanyas/mimo (the reference, Python/NumPy/SciPy), function by
function, keeping the reference's own contraction strings and evaluation order so that it can be
(a) checked against golden vectors produced by importing the reference itself
(tests/golden/make_golden.py -> tests/golden/*.npz; checked in tests/test_oracle_golden.py) and
(b) used as the checker for the HIP path and as the `cpu_baseline` of bench.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Parity status: PINNED against fixtures generated from the reference in the build container
(the reference ships no tests / golden vectors of its own — SURVEY.md §4, §8(c)).

Every function cites the reference lines it follows (paths relative to the reference root).
All arithmetic is float64; per-component tables are (K, N), K-major, like the reference.
"""
import numpy as np
from scipy.special import logsumexp, digamma, gammaln, betaln, multigammaln
from scipy import linalg as sla

LOG2PI = np.log(2.0 * np.pi)


# ==========================================================================================
# utilities — mimo/utils
# ==========================================================================================
def one_hot(z, K):
    """mimo/utils/data.py:160-169 — dense (K, N) 0/1 float64 table."""
    z = np.atleast_1d(z).astype(int)
    assert np.all(z >= 0) and np.all(z < K)
    N = z.size
    zoh = np.zeros((K, N))
    zoh[z.ravel(), np.arange(N)] = 1
    return zoh


def sample_discrete_from_log(p_log, u, dtype=np.int32):
    """mimo/utils/stats.py:8-21 with axis=0; `u` (shape (N,) or (1,N)) stands for the single
    npr.random(size=(1, N)) call the reference makes."""
    lognorms = logsumexp(p_log, axis=0)
    cumvals = np.exp(p_log - np.expand_dims(lognorms, 0)).cumsum(0)
    randvals = np.reshape(u, (1, -1)) * np.reshape(cumvals[-1], (1, -1))
    return np.sum(randvals > cumvals, axis=0, dtype=dtype)


# ---- counter-based uniforms of the in-kernel generator (not in the reference) ----------------
def philox_uniforms(seed, rows, sweep):
    """Philox4x32-10, key = seed (lo, hi), counter = (row_lo, row_hi, sweep_lo, sweep_hi);
    u = ((x0 >> 5) * 2^26 + (x1 >> 6)) / 2^53.  Mirrors mimo_kernels.hip:philox_uniform."""
    rows = np.asarray(rows, dtype=np.uint64)
    M32 = np.uint64(0xFFFFFFFF)
    c0 = rows & M32
    c1 = rows >> np.uint64(32)
    c2 = np.full_like(rows, np.uint64(sweep) & M32)
    c3 = np.full_like(rows, np.uint64(sweep) >> np.uint64(32))
    k0 = np.uint64(seed) & M32
    k1 = np.uint64(seed) >> np.uint64(32)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ k0
        n1 = p1 & M32
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ k1
        n3 = p0 & M32
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & M32
        k1 = (k1 + np.uint64(0xBB67AE85)) & M32
    hi = (c0 >> np.uint64(5)).astype(np.float64)
    lo = (c1 >> np.uint64(6)).astype(np.float64)
    return (hi * 67108864.0 + lo) * (1.0 / 9007199254740992.0)


# ==========================================================================================
# gating — mimo/distributions/{categorical,dirichlet}.py, bayesian.py:36-179
# ==========================================================================================
def categorical_statistics(labels, K):
    """categorical.py:35-37 — bincount."""
    return np.bincount(labels, minlength=K)


def categorical_weighted_statistics(weights):
    """categorical.py:41-43 — sum_n r_kn."""
    return np.sum(np.atleast_2d(weights), axis=1)


def categorical_log_likelihood(probs, x):
    """categorical.py:51-59 for integer x without NaNs."""
    return np.log(probs)[list(x)]


def dirichlet_expected_statistics(alphas):
    """dirichlet.py:85-87."""
    return digamma(alphas) - digamma(np.sum(alphas))


def dirichlet_log_partition(alphas):
    """dirichlet.py:78-79."""
    return np.sum(gammaln(alphas)) - gammaln(np.sum(alphas))


def dirichlet_vlb(post_alphas, prior_alphas):
    """bayesian.py:93-96 with dirichlet.py:89-97: entropy(q) - cross_entropy(q, p)."""
    stats = dirichlet_expected_statistics(post_alphas)
    entropy = dirichlet_log_partition(post_alphas) - (post_alphas - 1.).dot(stats)
    cross = dirichlet_log_partition(prior_alphas) - (prior_alphas - 1.).dot(stats)
    return entropy - cross


def stick_acc_counts(counts):
    """bayesian.py:143,154 — reverse cumulative counts, last entry 0."""
    return np.hstack((np.cumsum(counts[::-1])[-2::-1], 0))


def stick_expected_statistics(gammas, deltas):
    """dirichlet.py:201-204."""
    E_log_stick = digamma(gammas) - digamma(gammas + deltas)
    E_log_rest = digamma(deltas) - digamma(gammas + deltas)
    return E_log_stick, E_log_rest


def stick_log_partition(gammas, deltas):
    """dirichlet.py:195-196."""
    return np.sum(betaln(gammas, deltas))


def stick_vlb(post, prior):
    """bayesian.py:173-176 with dirichlet.py:206-214."""
    s0, s1 = stick_expected_statistics(*post)
    entropy = stick_log_partition(*post) - ((post[0] - 1.).dot(s0) + (post[1] - 1.).dot(s1))
    cross = stick_log_partition(*prior) - ((prior[0] - 1.).dot(s0) + (prior[1] - 1.).dot(s1))
    return entropy - cross


def stick_probs_from_betas(betas_head):
    """dirichlet.py:177-186 — pi_k = beta_k prod_{j<k}(1-beta_j) with the last beta forced to 1."""
    betas = np.hstack((betas_head, 1.))
    probs = np.zeros((betas.shape[0],))
    probs[0] = betas[0]
    probs[1:] = betas[1:] * np.cumprod(1.0 - betas[:-1])
    return probs


def stick_mean(gammas, deltas):
    """TruncatedStickBreaking.mean (dirichlet.py:177-186): stick probabilities of the mean betas."""
    return stick_probs_from_betas((gammas / (gammas + deltas))[:-1])


def tied_nw_nat_to_std(a, b, c, d, D):
    """TiedNormalWisharts.nat_to_std (composite.py:273-283): the Wishart block is pooled over k."""
    mus = a / b[:, None]
    psi = np.linalg.inv(np.mean(c - np.einsum('k,kd,kl->kdl', b, mus, mus), axis=0))
    nu = np.mean(d + D)
    return mus, b, np.array(len(b) * [psi]), np.array(len(b) * [nu])


def tied_mnw_nat_to_std(a, b, c, d, row_dim, column_dim):
    """TiedMatrixNormalWisharts.nat_to_std (composite.py:798-808)."""
    Ms = np.einsum('kdl,klh->kdh', a, np.linalg.inv(b))
    psi = np.linalg.inv(np.mean(c - np.einsum('kdl,klm,khm->kdh', Ms, b, Ms), axis=0))
    nu = np.mean(d + row_dim + 1 - column_dim)
    return Ms, b, np.array(len(b) * [psi]), np.array(len(b) * [nu])


def gating_expected_log(gating_type, post):
    """gmm.py:246-252 — E[log pi_k] for Dirichlet, or E_log_stick_k + sum_{j<k} E_log_rest_j."""
    if gating_type == 'dirichlet':
        return dirichlet_expected_statistics(post)
    log_stick, log_rest = stick_expected_statistics(*post)
    return log_stick + np.hstack((0, np.cumsum(log_rest)[:-1]))


# ==========================================================================================
# Gaussian components — mimo/distributions/gaussian.py:377-542
# ==========================================================================================
def gauss_log_partition(mus, lmbdas):
    """gaussian.py:352-354 per component (lmbda_chol = upper Cholesky, :298-301)."""
    out = np.zeros(mus.shape[0])
    for k in range(mus.shape[0]):
        chol = sla.cholesky(lmbdas[k], lower=False)
        out[k] = 0.5 * np.einsum('d,dl,l->', mus[k], lmbdas[k], mus[k]) - np.sum(np.log(np.diag(chol)))
    return out


def gauss_log_base(K, D):
    """gaussian.py:69-74, 459-464."""
    return np.log(np.power(2. * np.pi, - D / 2.)) * np.ones(K)


def gauss_log_likelihood(x, mus, lmbdas):
    """gaussian.py:510-521.  Rows that hold a NaN: nan_to_num (on a copy here; the reference edits the caller's array
    in place), then the data-dependent part of those rows is set to 0 — they keep the normaliser-only value."""
    K, D = mus.shape
    bads = np.isnan(np.atleast_2d(x)).any(axis=1)
    x = np.nan_to_num(np.array(x, dtype=float)).reshape((-1, D))
    log_lik = np.einsum('kd,kdl,nl->kn', mus, lmbdas, x, optimize=True)\
        - 0.5 * np.einsum('nd,kdl,nl->kn', x, lmbdas, x, optimize=True)
    log_lik[:, bads] = 0.
    log_lik += - np.expand_dims(gauss_log_partition(mus, lmbdas), axis=1)\
        + np.expand_dims(gauss_log_base(K, D), axis=1)
    return log_lik


def gauss_weighted_statistics(data, weights):
    """gaussian.py:491-502 -> (xk, nk, xxTk, nk); rows that hold a NaN are dropped together with their weights."""
    idx = ~np.isnan(data).any(axis=1)
    data, weights = data[idx], weights[:, idx]
    xk = np.einsum('kn,nd->kd', weights, data, optimize=True)
    xxTk = np.einsum('nd,kn,nl->kdl', data, weights, data, optimize=True)
    nk = np.sum(weights, axis=1)
    return xk, nk, xxTk, nk


def gauss_statistics_unfolded(data, K):
    """gaussian.py:466-485 with fold=False: per-datum statistics replicated K times."""
    x = np.einsum('nd->nd', data, optimize=True)
    xxT = np.einsum('nd,nl->ndl', data, data, optimize=True)
    n = np.ones((data.shape[0],))
    xk = np.array([x for _ in range(K)])
    xxTk = np.array([xxT for _ in range(K)])
    nk = np.array([n for _ in range(K)])
    return xk, nk, xxTk, nk


# ==========================================================================================
# diagonal-precision Gaussians / Normal-Gamma blocks — mimo/distributions/gaussian.py:697-878,
# composite.py:286-547, bayesian.py:343-500
# ==========================================================================================
def diag_gauss_log_likelihood(x, mus, lmbdas_diags):
    """StackedGaussiansWithDiagonalPrecision.log_likelihood (gaussian.py:817-832) with
    log_partition (gaussian.py:678-680) and log_base (gaussian.py:69-74) per component."""
    K, D = mus.shape
    lmbdas = np.array([np.diag(l) for l in lmbdas_diags])
    log_lik = np.einsum('kd,kdl,nl->kn', mus, lmbdas, x, optimize=True)\
        - 0.5 * np.einsum('nd,kdl,nl->kn', x, lmbdas, x, optimize=True)
    log_partition = np.array([0.5 * np.einsum('d,dl,l->', mus[k], lmbdas[k], mus[k])
                              - np.sum(np.log(np.sqrt(lmbdas_diags[k]))) for k in range(K)])
    return log_lik - log_partition[:, None] - 0.5 * D * LOG2PI


def diag_gauss_weighted_statistics(data, weights):
    """gaussian.py:802-815 -> (xk (K,D), ndk (K,D), ndk, xxk (K,D))."""
    K, D = weights.shape[0], data.shape[1]
    xk = np.einsum('kn,nd->kd', weights, data)
    xxk = np.einsum('nd,kn,nd->kd', data, weights, data)
    ndk = np.broadcast_to(np.sum(weights, axis=1, keepdims=True), (K, D))
    return xk, ndk, ndk, xxk


def ng_std_to_nat(mus, kappas, alphas, betas):
    """composite.py:314-329."""
    return kappas * mus, kappas, 2. * alphas - 1., 2. * betas + kappas * mus**2


def ng_nat_to_std(a, b, c, d):
    """composite.py:331-337."""
    mus = a / b
    return mus, b, 0.5 * (c + 1.), 0.5 * (d - b * mus**2)


def tied_ng_nat_to_std(a, b, c, d):
    """composite.py:536-547: Gamma factor pooled (mean over k)."""
    K = a.shape[0]
    mus = a / b
    alphas = np.mean(0.5 * (c + 1.), axis=0)
    betas = np.mean(0.5 * (d - b * mus**2), axis=0)
    return mus, b, np.array(K * [alphas]), np.array(K * [betas])


def ng_expected_statistics(mus, kappas, alphas, betas):
    """composite.py:371-382."""
    E_lmbdas_mu = alphas / betas * mus
    return (E_lmbdas_mu, - 0.5 * (1. / kappas + mus * E_lmbdas_mu),
            0.5 * (digamma(alphas) - np.log(betas)), - 0.5 * (alphas / betas))


def ng_log_partition(mus, kappas, alphas, betas):
    """composite.py:360-363 + gamma.py:91-92, per block."""
    return - 0.5 * np.sum(np.log(kappas), axis=1) + np.sum(gammaln(alphas) - alphas * np.log(betas), axis=1)


def ng_vlb(post, prior):
    """entropy - cross_entropy (bayesian.py:401-404, composite.py:384-404), per block."""
    D = post[0].shape[1]
    E = ng_expected_statistics(*post)
    inner = lambda nat: sum(np.einsum('kd,kd->k', n, e) for n, e in zip(nat, E))
    log_base = - 0.5 * D * LOG2PI
    ent = ng_log_partition(*post) - log_base - inner(ng_std_to_nat(*post))
    cross = ng_log_partition(*prior) - log_base - inner(ng_std_to_nat(*prior))
    return ent - cross


def stacked_ng_update(prior, stats, post_gamma, tied=False):
    """posterior.nat_param = prior.nat_param + stats as the reference OBSERVABLY performs it
    (bayesian.py:385-392): nat_to_std yields four arrays, but the stacked setters of alphas / betas store
    them in attributes nothing reads (composite.py:472-484), so (alphas, betas) stay `post_gamma`."""
    nat = [p + s for p, s in zip(ng_std_to_nat(*prior), stats)]
    mus, kappas, _, _ = (tied_ng_nat_to_std if tied else ng_nat_to_std)(*nat)
    return mus, kappas, post_gamma[0], post_gamma[1]


def diag_gauss_ng_expected_log_likelihood(x, post):
    """StackedGaussiansWithNormalGammas.expected_log_likelihood (bayesian.py:441-455) with the unfolded
    statistics [x, 1, 1, x^2] of gaussian.py:784-800."""
    K, D = post[0].shape
    E = ng_expected_statistics(*post)
    xk = np.array([x for _ in range(K)])
    xxk = np.array([x * x for _ in range(K)])
    ndk = np.ones((K, x.shape[0], D))
    return - 0.5 * D * LOG2PI + np.einsum('kd,knd->kn', E[0], xk) + np.einsum('kd,knd->kn', E[1], ndk)\
        + np.einsum('kd,knd->kn', E[2], ndk) + np.einsum('kd,knd->kn', E[3], xxk)


# ==========================================================================================
# Wishart / Normal-Wishart — mimo/distributions/wishart.py, composite.py:19-256
# ==========================================================================================
def wishart_log_partition(psi, nu):
    """wishart.py:129-132."""
    D = psi.shape[0]
    return 0.5 * nu * D * np.log(2) + multigammaln(nu / 2., D)\
        + nu * np.sum(np.log(np.diag(np.linalg.cholesky(psi))))


def wishart_rvs(psi, nu, npr):
    """wishart.py:72-92 — Bartlett sampler, same RNG call order (normal(n_tril), then D chisquare)."""
    D = psi.shape[0]
    n_tril = D * (D - 1) // 2
    covariances = npr.normal(size=n_tril).reshape((n_tril,))
    variances = (np.r_[[npr.chisquare(nu - (i + 1) + 1, size=1) ** 0.5 for i in range(D)]].reshape((D,)).T)
    A = np.zeros((D, D))
    A[np.tril_indices(D, k=-1)] = covariances
    A[np.diag_indices(D)] = variances
    T = np.dot(np.linalg.cholesky(psi), A)
    return np.dot(T, T.T)


def nw_std_to_nat(mu, kappa, psi, nu):
    """composite.py:50-65."""
    D = mu.shape[0]
    return kappa * mu, kappa, np.linalg.inv(psi) + kappa * np.outer(mu, mu), nu - D


def nw_nat_to_std(a, b, c, d):
    """composite.py:67-72."""
    D = a.shape[0]
    mu = a / b
    return mu, b, np.linalg.inv(c - b * np.outer(mu, mu)), d + D


def nw_expected_statistics(mu, kappa, psi, nu):
    """composite.py:106-118."""
    D = mu.shape[0]
    E_lmbda_mu = nu * psi @ mu
    E_muT_lmbda_mu = - 0.5 * (D / kappa + mu.dot(E_lmbda_mu))
    E_lmbda = - 0.5 * (nu * psi)
    E_logdet_lmbda = 0.5 * (np.sum(digamma((nu - np.arange(D)) / 2.))
                            + D * np.log(2.) + 2. * np.sum(np.log(np.diag(np.linalg.cholesky(psi)))))
    return E_lmbda_mu, E_muT_lmbda_mu, E_lmbda, E_logdet_lmbda


def nw_log_partition(mu, kappa, psi, nu):
    """composite.py:95-98."""
    D = mu.shape[0]
    return - 0.5 * D * np.log(kappa) + wishart_log_partition(psi, nu)


def nw_log_base(D):
    """composite.py:88-93: gaussian.base * wishart.base."""
    return np.log(np.power(2. * np.pi, - D / 2.) * 1.)


def nw_vlb(post, prior):
    """bayesian.py:240-243 with composite.py:120-134: entropy(q) - cross_entropy(q, p), one component."""
    D = post[0].shape[0]
    stats = nw_expected_statistics(*post)

    def inner(nat):
        return np.dot(nat[0], stats[0]) + nat[1] * stats[1] + np.tensordot(nat[2], stats[2]) + nat[3] * stats[3]
    entropy = nw_log_partition(*post) - nw_log_base(D) - inner(nw_std_to_nat(*post))
    cross = nw_log_partition(*prior) - nw_log_base(D) - inner(nw_std_to_nat(*prior))
    return entropy - cross


def nw_rvs(mu, kappa, psi, nu, npr):
    """composite.py:82-86 + gaussian.py:311-313: Lambda ~ W(psi, nu), mu ~ N(m, (kappa Lambda)^-1)."""
    D = mu.shape[0]
    lmbda = wishart_rvs(psi, nu, npr)
    chol = sla.cholesky(kappa * lmbda, lower=False)
    chol_inv = sla.inv(chol)
    m = mu + npr.normal(size=D).dot(chol_inv.T)
    return m, lmbda


def stacked_nw_update(prior, stats):
    """bayesian.py:217-230 + composite.py:170-184: posterior.nat = prior.nat + stats, per component.
    prior = (mus, kappas, psis, nus) stacked; stats = (xk, nk, xxTk, nk)."""
    mus, kappas, psis, nus = prior
    out = [[], [], [], []]
    for k in range(mus.shape[0]):
        nat = nw_std_to_nat(mus[k], kappas[k], psis[k], nus[k])
        nat = tuple(n + s for n, s in zip(nat, (stats[0][k], stats[1][k], stats[2][k], stats[3][k])))
        for o, v in zip(out, nw_nat_to_std(*nat)):
            o.append(v)
    return tuple(np.stack(o, axis=0) for o in out)


def gauss_nw_expected_log_likelihood(x, post, chunk=1024):
    """bayesian.py:287-301 (stacked): <E_q[eta_k], t(x_n)> with statistics(fold=False) —
    the (K,N,D,D) replication of gaussian.py:481-485 is kept, N is processed in chunks."""
    mus, kappas, psis, nus = post
    K, D = mus.shape
    nat = [np.stack(v, axis=0) for v in zip(*[nw_expected_statistics(mus[k], kappas[k], psis[k], nus[k])
                                               for k in range(K)])]
    log_base = gauss_log_base(K, D)
    out = np.empty((K, x.shape[0]))
    for s in range(0, x.shape[0], chunk):
        stats = gauss_statistics_unfolded(x[s:s + chunk], K)
        out[:, s:s + chunk] = np.expand_dims(log_base, axis=1)\
            + np.einsum('kd,knd->kn', nat[0], stats[0])\
            + np.einsum('k,kn->kn', nat[1], stats[1])\
            + np.einsum('kdl,kndl->kn', nat[2], stats[2])\
            + np.einsum('k,kn->kn', nat[3], stats[3])
    return out


# ==========================================================================================
# linear-Gaussian experts — mimo/distributions/lingauss.py:187-367
# ==========================================================================================
def lingauss_predict(x, As, affine=True):
    """lingauss.py:251-257."""
    if affine:
        A, b = As[:, :, :-1], As[:, :, -1]
        return np.einsum('kdl,...l->k...d', A, x, optimize=True) + b[:, None, :]
    return np.einsum('kdl,...l->k...d', As, x, optimize=True)


def lingauss_log_partition(x, As, lmbdas, affine=True):
    """lingauss.py:166-169, 327-328 — per component, depends on x."""
    K = As.shape[0]
    out = np.empty((K, x.shape[0]))
    mu = lingauss_predict(x, As, affine)
    for k in range(K):
        chol = sla.cholesky(lmbdas[k], lower=False)
        out[k] = 0.5 * np.einsum('nd,dl,nl->n', mu[k], lmbdas[k], mu[k]) - np.sum(np.log(np.diag(chol)))
    return out


def lingauss_log_likelihood(x, y, As, lmbdas, affine=True, x_already_cleaned=False):
    """lingauss.py:330-345.  Rows with a NaN: x and y go through nan_to_num element by element (on copies here; the
    reference edits the caller's arrays in place), and the data part is set to 0 only where x AND y hold a NaN (:332-333,
    :344).  `x_already_cleaned`: the call inside log_complete_likelihood (ilr.py:71-75), where the input density has run
    first and nan_to_num'ed x IN PLACE — the experts' density then finds no NaN in x and zeroes no row."""
    K, dy = lmbdas.shape[0], lmbdas.shape[1]
    bx = np.isnan(np.atleast_2d(x)).any(axis=1)
    by = np.isnan(np.atleast_2d(y)).any(axis=1)
    bads = np.zeros_like(by) if x_already_cleaned else np.logical_and(bx, by)
    x = np.nan_to_num(np.array(x, dtype=float)).reshape((-1, As.shape[2] - (1 if affine else 0)))
    y = np.nan_to_num(np.array(y, dtype=float)).reshape((-1, dy))
    mu = lingauss_predict(x, As, affine)
    log_lik = np.einsum('knd,kdl,nl->kn', mu, lmbdas, y, optimize=True)\
        - 0.5 * np.einsum('nd,kdl,nl->kn', y, lmbdas, y, optimize=True)
    log_lik[:, bads] = 0.
    log_lik += - lingauss_log_partition(x, As, lmbdas, affine)\
        + np.expand_dims(np.log(np.power(2. * np.pi, - dy / 2.)) * np.ones(K), axis=1)
    return log_lik


def lingauss_weighted_statistics(x, y, weights, affine=True):
    """lingauss.py:306-322 -> (yxTk, xxTk, yyTk, nk); rows with a NaN in x or y are dropped together with their weights (:308-310)."""
    idx = np.logical_and(~np.isnan(x).any(axis=1), ~np.isnan(y).any(axis=1))
    x, y, weights = x[idx], y[idx], weights[:, idx]
    if affine:
        x = np.hstack((x, np.ones((x.shape[0], 1))))
    contract = 'nd,kn,nl->kdl'
    yxTk = np.einsum(contract, y, weights, x, optimize=True)
    xxTk = np.einsum(contract, x, weights, x, optimize=True)
    yyTk = np.einsum(contract, y, weights, y, optimize=True)
    nk = np.sum(weights, axis=1)
    return yxTk, xxTk, yyTk, nk


def lingauss_statistics_unfolded(x, y, K, affine=True):
    """lingauss.py:275-300 with fold=False."""
    if affine:
        x = np.hstack((x, np.ones((x.shape[0], 1))))
    contract = 'nd,nl->ndl'
    n = np.ones((y.shape[0],))
    yxT = np.einsum(contract, y, x, optimize=True)
    xxT = np.einsum(contract, x, x, optimize=True)
    yyT = np.einsum(contract, y, y, optimize=True)
    return (np.array([yxT for _ in range(K)]), np.array([xxT for _ in range(K)]),
            np.array([yyT for _ in range(K)]), np.array([n for _ in range(K)]))


# ---- Matrix-Normal-Wishart — composite.py:550-783, matrix.py:10-175 --------------------------
def mnw_std_to_nat(M, Kmat, psi, nu):
    """composite.py:577-592."""
    dy, dx = M.shape
    return M @ Kmat, Kmat, np.linalg.inv(psi) + M @ Kmat @ M.T, nu - dy - 1. + dx


def mnw_nat_to_std(a, b, c, d):
    """composite.py:594-599."""
    dy, dx = a.shape
    M = a @ np.linalg.inv(b)
    return M, b, np.linalg.inv(c - M @ b @ M.T), d + dy + 1. - dx


def mnw_expected_statistics(M, Kmat, psi, nu):
    """composite.py:635-647."""
    dy = M.shape[0]
    E_Lmbda_A = nu * psi @ M
    E_AT_Lmbda_A = - 0.5 * (dy * np.linalg.inv(Kmat) + M.T.dot(E_Lmbda_A))
    E_lmbda = - 0.5 * (nu * psi)
    E_logdet_lmbda = 0.5 * (np.sum(digamma((nu - np.arange(dy)) / 2.))
                            + dy * np.log(2.) + 2. * np.sum(np.log(np.diag(np.linalg.cholesky(psi)))))
    return E_Lmbda_A, E_AT_Lmbda_A, E_lmbda, E_logdet_lmbda


def mnw_log_partition(M, Kmat, psi, nu):
    """composite.py:622-625."""
    dy = M.shape[0]
    return - 0.5 * dy * np.linalg.slogdet(Kmat)[1] + wishart_log_partition(psi, nu)


def mnw_log_base(dy, dx):
    """composite.py:615-620 with matrix.py:127-132."""
    return np.log(np.power(2. * np.pi, - dy * dx / 2.) * 1.)


def mnw_vlb(post, prior):
    """bayesian.py:854-857 with composite.py:649-663."""
    dy, dx = post[0].shape
    stats = mnw_expected_statistics(*post)

    def inner(nat):
        return (np.tensordot(nat[0], stats[0]) + np.tensordot(nat[1], stats[1])
                + np.tensordot(nat[2], stats[2]) + nat[3] * stats[3])
    entropy = mnw_log_partition(*post) - mnw_log_base(dy, dx) - inner(mnw_std_to_nat(*post))
    cross = mnw_log_partition(*prior) - mnw_log_base(dy, dx) - inner(mnw_std_to_nat(*prior))
    return entropy - cross


def mnw_rvs(M, Kmat, psi, nu, npr):
    """composite.py:607-611 + matrix.py:122-125: Lambda ~ W(psi,nu); vec_F(A) ~ N(vec_F(M), kron(K,Lambda)^-1)."""
    dy, dx = M.shape
    lmbda = wishart_rvs(psi, nu, npr)
    chol = sla.cholesky(np.kron(Kmat, lmbda), lower=False)
    chol_inv = sla.inv(chol)
    aux = npr.normal(size=dy * dx).dot(chol_inv.T)
    return M + np.reshape(aux, (dy, dx), order='F'), lmbda


def stacked_mnw_update(prior, stats):
    """bayesian.py:831-844: posterior.nat = prior.nat + (yxT, xxT, yyT, n)."""
    Ms, Ks, psis, nus = prior
    out = [[], [], [], []]
    for k in range(Ms.shape[0]):
        nat = mnw_std_to_nat(Ms[k], Ks[k], psis[k], nus[k])
        nat = tuple(n + s for n, s in zip(nat, (stats[0][k], stats[1][k], stats[2][k], stats[3][k])))
        for o, v in zip(out, mnw_nat_to_std(*nat)):
            o.append(v)
    return tuple(np.stack(o, axis=0) for o in out)


def lingauss_mnw_expected_log_likelihood(x, y, post, affine=True, chunk=1024):
    """bayesian.py:933-947 (stacked)."""
    Ms, Ks, psis, nus = post
    K, dy = Ms.shape[0], Ms.shape[1]
    nat = [np.stack(v, axis=0) for v in zip(*[mnw_expected_statistics(Ms[k], Ks[k], psis[k], nus[k])
                                               for k in range(K)])]
    log_base = np.log(np.power(2. * np.pi, - dy / 2.)) * np.ones(K)
    out = np.empty((K, x.shape[0]))
    for s in range(0, x.shape[0], chunk):
        stats = lingauss_statistics_unfolded(x[s:s + chunk], y[s:s + chunk], K, affine)
        out[:, s:s + chunk] = np.expand_dims(log_base, axis=1)\
            + np.einsum('kdl,kndl->kn', nat[0], stats[0])\
            + np.einsum('kdl,kndl->kn', nat[1], stats[1])\
            + np.einsum('kdl,kndl->kn', nat[2], stats[2])\
            + np.einsum('k,kn->kn', nat[3], stats[3])
    return out


# ==========================================================================================
# mixture level — mimo/mixtures/gmm.py, ilr.py
# ==========================================================================================
def responsibilities(log_lik):
    """gmm.py:72-75, 256-259: exp(l - logsumexp_k l)."""
    return np.exp(log_lik - logsumexp(log_lik, axis=0, keepdims=True))


def gmm_log_complete_likelihood(x, mus, lmbdas, probs):
    """gmm.py:67-70."""
    return gauss_log_likelihood(x, mus, lmbdas) + np.expand_dims(np.log(probs), axis=1)


def gmm_expected_log_complete_likelihood(x, post, gating_type, gating_post, chunk=1024):
    """gmm.py:244-254."""
    return gauss_nw_expected_log_likelihood(x, post, chunk)\
        + np.expand_dims(gating_expected_log(gating_type, gating_post), axis=1)


def vlb_labels(resp, gating_type, gating_post):
    """gmm.py:341-356 / ilr.py:299-314."""
    vlb = 0.
    if gating_type == 'dirichlet':
        vlb += np.sum(resp * np.expand_dims(dirichlet_expected_statistics(gating_post), axis=1))
    else:
        acc_resp = np.vstack((np.cumsum(resp[::-1, :], axis=0)[-2::-1, :], np.zeros((1, resp.shape[-1]))))
        E_log_stick, E_log_rest = stick_expected_statistics(*gating_post)
        vlb += np.sum(resp * np.expand_dims(E_log_stick, axis=1) + acc_resp * np.expand_dims(E_log_rest, axis=1))
    with np.errstate(invalid='ignore', divide='ignore'):
        vlb -= np.nansum(resp * np.log(resp))
    return vlb


def gating_vlb(gating_type, post, prior):
    return dirichlet_vlb(post, prior) if gating_type == 'dirichlet' else stick_vlb(post, prior)


def gating_update(gating_type, prior, counts):
    """bayesian.py:70-83 (Dirichlet: alpha0 + counts) / :140-159 (stick-breaking)."""
    if gating_type == 'dirichlet':
        return prior + counts
    return prior[0] + counts, prior[1] + stick_acc_counts(counts)


def gmm_vi_iteration(x, prior, gprior, gating_type, resp):
    """One pass of the loop body of gmm.py:275-285 without the (numerically irrelevant) rvs:
    meanfield_update_parameters -> expected_responsibilities -> variational_lowerbound."""
    stats = gauss_weighted_statistics(x, resp)
    post = stacked_nw_update(prior, stats)
    gpost = gating_update(gating_type, gprior, categorical_weighted_statistics(resp))
    comp_ll = gauss_nw_expected_log_likelihood(x, post)
    log_lik = comp_ll + np.expand_dims(gating_expected_log(gating_type, gpost), axis=1)
    new_resp = responsibilities(log_lik)
    vlb = gating_vlb(gating_type, gpost, gprior)
    vlb += np.sum([nw_vlb(tuple(p[k] for p in post), tuple(p[k] for p in prior)) for k in range(post[0].shape[0])])
    vlb += np.sum(new_resp * comp_ll)                      # gmm.py:338-339
    vlb += vlb_labels(new_resp, gating_type, gpost)        # gmm.py:341-356
    return post, gpost, new_resp, vlb


def ilr_log_complete_likelihood(x, y, mus, lmbdas, As, lmbdas_y, probs, affine=True):
    """ilr.py:71-75 (the input density runs first and cleans x in place: see lingauss_log_likelihood)."""
    return gauss_log_likelihood(x, mus, lmbdas) + lingauss_log_likelihood(x, y, As, lmbdas_y, affine, x_already_cleaned=True)\
        + np.expand_dims(np.log(probs), axis=1)


def ilr_expected_log_complete_likelihood(x, y, bpost, mpost, gating_type, gating_post, affine=True):
    """ilr.py:178-189."""
    return gauss_nw_expected_log_likelihood(x, bpost) + lingauss_mnw_expected_log_likelihood(x, y, mpost, affine)\
        + np.expand_dims(gating_expected_log(gating_type, gating_post), axis=1)


def ilr_vi_iteration(x, y, bprior, mprior, gprior, gating_type, resp, affine=True):
    """Loop body of ilr.py:216-226 without the rvs."""
    K = resp.shape[0]
    bpost = stacked_nw_update(bprior, gauss_weighted_statistics(x, resp))
    mpost = stacked_mnw_update(mprior, lingauss_weighted_statistics(x, y, resp, affine))
    gpost = gating_update(gating_type, gprior, categorical_weighted_statistics(resp))
    b_ll = gauss_nw_expected_log_likelihood(x, bpost)
    m_ll = lingauss_mnw_expected_log_likelihood(x, y, mpost, affine)
    log_lik = b_ll + m_ll + np.expand_dims(gating_expected_log(gating_type, gpost), axis=1)
    new_resp = responsibilities(log_lik)
    vlb = gating_vlb(gating_type, gpost, gprior)
    vlb += np.sum([nw_vlb(tuple(p[k] for p in bpost), tuple(p[k] for p in bprior)) for k in range(K)])
    vlb += np.sum([mnw_vlb(tuple(p[k] for p in mpost), tuple(p[k] for p in mprior)) for k in range(K)])
    vlb += np.sum(new_resp * b_ll) + np.sum(new_resp * m_ll)   # ilr.py:293-297
    vlb += vlb_labels(new_resp, gating_type, gpost)
    return bpost, mpost, gpost, new_resp, vlb


# ---- Gibbs sweep (gmm.py:220-223) ------------------------------------------------------------
def gmm_gibbs_sweep(x, prior, gprior, gating_type, labels, npr, u=None):
    """resample_components -> resample_gating -> resample_labels with the reference's RNG call order
    (per component: Wishart normal(n_tril), D x chisquare, normal(D); then Dirichlet / Beta; then
    random((1,N)) unless `u` is supplied)."""
    K = prior[0].shape[0]
    weights = one_hot(labels, K)
    post = stacked_nw_update(prior, gauss_weighted_statistics(x, weights))
    draws = [nw_rvs(post[0][k], post[1][k], post[2][k], post[3][k], npr) for k in range(K)]
    mus = np.stack([d[0] for d in draws])
    lmbdas = np.stack([d[1] for d in draws])
    counts = categorical_statistics(np.asarray(labels).astype(int), K)
    if gating_type == 'dirichlet':
        gpost = gprior + counts
        probs = np.clip(npr.dirichlet(gpost), np.spacing(1.), np.inf)       # bayesian.py:70-75
    else:
        gpost = (gprior[0] + counts, gprior[1] + stick_acc_counts(counts))
        probs = stick_probs_from_betas(npr.beta(gpost[0][:-1], gpost[1][:-1]))  # dirichlet.py:177-186
    log_prob = gmm_log_complete_likelihood(x, mus, lmbdas, probs)
    if u is None:
        u = npr.random(size=(1, x.shape[0]))
    new_labels = sample_discrete_from_log(log_prob, u)
    return post, gpost, mus, lmbdas, probs, new_labels


# ==========================================================================================
# canonical form — derived, used to cross-check the engine's inputs
# ==========================================================================================
def canonical_eval(z, c, b, W):
    """l[k,n] = c_k + b_k.z_n - 1/2 z_n' W_k z_n."""
    return c[:, None] + b @ z.T - 0.5 * np.einsum('nd,kde,ne->kn', z, W, z, optimize=True)


def packed_stats(z, weights):
    """(n_k, sum r z, sum r z z') for z rows — the engine's packed output, computed naively."""
    return weights.sum(1), weights @ z, np.einsum('kn,nd,ne->kde', weights, z, z, optimize=True)


# ---------------------------------------------------------------------------------------------
# posterior-predictive path
# ---------------------------------------------------------------------------------------------
def stacked_mvn_logpdf(xs, mus, lmbdas):
    """mimo/utils/stats.py:53-66 (stacked_multivariate_gaussian_loglik) -> (K, N)."""
    d = mus.shape[-1]
    xc = xs[:, None, :] - mus[None, :, :]
    log_exps = - 0.5 * np.einsum('nkd,kdl,nkl->kn', xc, lmbdas, xc)
    log_norms = - 0.5 * d * np.log(2. * np.pi) + 0.5 * np.linalg.slogdet(lmbdas)[1]
    return log_norms[:, None] + log_exps


def nw_posterior_predictive_gaussian(post):
    """bayesian.py:303-309: moment-matched Gaussian of the Normal-Wishart posterior predictive."""
    mus, kappas, psis, nus = post
    dfs = nus - mus.shape[-1] + 1
    cs = 1. + 1. / kappas
    return mus, np.einsum('k,kdl->kdl', dfs / cs, psis)


def mnw_posterior_predictive_gaussian(x, post, affine=True):
    """bayesian.py:949-962 -> mus (K,N,dy), lmbdas (K,N,dy,dy)."""
    Ms, Ks, psis, nus = post
    if affine:
        x = np.hstack((x, np.ones((len(x), 1))))
    dfs = nus - Ms.shape[1] + 1
    mus = np.einsum('kdl,nl->knd', Ms, x)
    cs = 1. + np.einsum('nd,kdl,nl->kn', x, np.linalg.inv(Ks), x)
    lmbdas = np.einsum('kdl,k,kn->kndl', psis, dfs, 1. / cs)
    return mus, lmbdas


def ilr_predictive_weights(x, bpost, gating_mean):
    """ilr.py:339-348 (dist='gaussian')."""
    log_pp = stacked_mvn_logpdf(x, *nw_posterior_predictive_gaussian(bpost))
    log_weight = np.log(gating_mean)[:, None] + log_pp
    return np.exp(log_weight - logsumexp(log_weight, axis=0, keepdims=True))


def mixture_moments(mus, covars, weights):
    """ilr.py:364-372."""
    mu = np.einsum('knd,kn->nd', mus, weights)
    covar = np.einsum('kndl,kn->ndl', covars + np.einsum('knd,knl->kndl', mus, mus), weights)\
        - np.einsum('nd,nl->ndl', mu, mu)
    return mu, covar


def ilr_meanfield_prediction(x, bpost, mpost, gating_mean, prediction='average', affine=True, y=None):
    """ilr.py:374-409 on already-transformed inputs (the scaling steps :384-389,411-414 are host-side
    affine maps).  With y: the predictive log-density the reference INTENDS at :405-409 — its own call
    raises for stacked models (stats.py:57 broadcasts (N,1,d) against (1,K,N,d)), so nlpd has no golden
    vector; the formula is log N(y_n; mu_kn, lmbda_kn^-1) per (k, n)."""
    weights = ilr_predictive_weights(x, bpost, gating_mean)
    mus, lmbdas = mnw_posterior_predictive_gaussian(x, mpost, affine)
    covars = np.linalg.inv(lmbdas)
    if prediction == 'mode':
        k = np.argmax(weights, axis=0)
        idx = (k, range(len(k)), ...)
        mu, covar = mus[idx], covars[idx]
    else:
        mu, covar = mixture_moments(mus, covars, weights)
    nlpd = None
    if y is not None:
        r = y[None, :, :] - mus
        d = y.shape[-1]
        log_pl = - 0.5 * np.einsum('knd,kndl,knl->kn', r, lmbdas, r) - 0.5 * d * np.log(2. * np.pi)\
            + 0.5 * np.linalg.slogdet(lmbdas)[1]
        nlpd = - logsumexp(log_pl + np.log(weights + np.finfo(np.float64).tiny), axis=0)
    return mu, covar, nlpd


def predict_canonical(z, c, b, W, M, Q, Cc, affine=True, mode='average', y=None, P=None, ld=None):
    """What mimo_predict computes, stated on its own arguments (include/mimo_hip.h)."""
    L = canonical_eval(z, c, b, W)
    weights = np.exp(L - logsumexp(L, axis=0, keepdims=True))
    xt = np.hstack((z, np.ones((len(z), 1)))) if affine else z
    mus = np.einsum('kdl,nl->knd', M, xt)
    cs = 1. + np.einsum('nd,kdl,nl->kn', xt, Q, xt)
    covars = np.einsum('kn,kdl->kndl', cs, Cc)
    if mode == 'mode':
        k = np.argmax(L, axis=0)
        idx = (k, range(len(k)), ...)
        mu, covar = mus[idx], covars[idx]
    else:
        mu, covar = mixture_moments(mus, covars, weights)
    nlpd = None
    if y is not None:
        r = y[None, :, :] - mus
        d = y.shape[-1]
        log_pl = - 0.5 * np.einsum('knd,kdl,knl->kn', r, P, r) / cs - 0.5 * d * np.log(2. * np.pi)\
            + 0.5 * (ld[:, None] - d * np.log(cs))
        nlpd = - logsumexp(log_pl + np.log(weights + np.finfo(np.float64).tiny), axis=0)
    return mu, covar, nlpd
