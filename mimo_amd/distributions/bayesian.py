"""Conjugate 'likelihood + prior + posterior' wrappers with the reference's method surface
(mimo/distributions/bayesian.py:36-179, 182-323, 796-985):

    resample / meanfield_update / meanfield_sgd / max_aposteriori / variational_lowerbound /
    expected_log_likelihood, attributes .prior .posterior .likelihood

Every update is  posterior.nat_param = prior.nat_param + stats  followed by a refresh of the
point-estimate likelihood.  `stats` is either computed from (data, weights) through the HIP engine
(reference-shaped calls) or handed over by the mixture drivers, which get the statistics of ALL
blocks from one fused pass over the data.
"""
import copy

import numpy as np

from mimo_amd.distributions.gating import Categorical
from mimo_amd.distributions.gaussian import (StackedGaussiansWithPrecision, TiedGaussiansWithPrecision,
                                             StackedGaussiansWithDiagonalPrecision,
                                             TiedGaussiansWithDiagonalPrecision)
from mimo_amd.distributions.lingauss import StackedLinearGaussiansWithPrecision, TiedLinearGaussiansWithPrecision


_STUDENTT_DEFECT = ("dist='studentt' has no observable behaviour in the reference for stacked blocks: "
                    "mimo/utils/stats.py:79 divides a (K,N) array by a (K,) vector and mimo/mixtures/ilr.py:355 "
                    "contracts a 4-D array with a 3-index subscript — both raise")


def stick_acc_counts(counts):
    """counts of all later sticks, last entry 0 (bayesian.py:143,154; Blei & Jordan 2006)."""
    return np.hstack((np.cumsum(counts[::-1])[-2::-1], 0))


class _CategoricalWrapper:

    def __init__(self, dim, prior, likelihood=None):
        self.dim = dim
        self.prior = prior
        self.posterior = copy.deepcopy(prior)
        if likelihood is not None:
            self.likelihood = likelihood
        else:
            self.likelihood = Categorical(dim=self.dim, probs=self.prior.rvs())

    def _counts(self, data, weights):
        return self.likelihood.statistics(data) if weights is None\
            else self.likelihood.weighted_statistics(data, weights)

    def variational_lowerbound(self):
        fused = getattr(self.posterior, 'entropy_minus_cross_entropy', None)
        if fused is not None:
            return fused(self.prior)
        return self.posterior.entropy() - self.posterior.cross_entropy(self.prior)

    def refresh_likelihood(self):
        """The draw meanfield_update(sample=True) ends with (bayesian.py:73-76 / 150-152 of the reference), as a
        step of its own: the mean-field drivers issue it AFTER the next data pass has been launched — the pass reads
        the posterior, never the drawn point estimate — so the host draws overlap the kernel."""
        self.likelihood.params = self.posterior.rvs()

    def expected_log_likelihood(self):
        return self.posterior.expected_statistics()


class CategoricalWithDirichlet(_CategoricalWrapper):
    """reference: bayesian.py:36-99"""

    def max_aposteriori(self, data, weights=None):
        self.posterior.nat_param = self.prior.nat_param + self._counts(data, weights)
        self.likelihood.params = self.posterior.mode()

    def resample(self, data, counts=None):
        counts = self.likelihood.statistics(data) if counts is None else counts
        self.posterior.nat_param = self.prior.nat_param + counts
        self.likelihood.params = np.clip(self.posterior.rvs(), np.spacing(1.), np.inf)

    def meanfield_update(self, data, weights=None, sample=True):
        self.posterior.nat_param = self.prior.nat_param + self._counts(data, weights)
        if sample:
            self.likelihood.params = self.posterior.rvs()

    def meanfield_sgd(self, data, weights, scale, step_size, sample=True):
        stats = self._counts(data, weights)
        self.posterior.nat_param = (1. - step_size) * self.posterior.nat_param\
            + step_size * (self.prior.nat_param + 1. / scale * stats)
        if sample:
            self.likelihood.params = self.posterior.rvs()

    def expected_log_gating(self):
        """E[log pi_k] per component (gmm.py:248-249)."""
        return self.posterior.expected_statistics()


class CategoricalWithStickBreaking(_CategoricalWrapper):
    """reference: bayesian.py:102-179"""

    def _set(self, counts):
        self.posterior.gammas = self.prior.gammas + counts
        self.posterior.deltas = self.prior.deltas + stick_acc_counts(counts)

    def max_aposteriori(self, data, weights=None):
        self._set(self._counts(data, weights))
        self.likelihood.params = self.posterior.mode()

    def resample(self, data, counts=None):
        self._set(self.likelihood.statistics(data) if counts is None else counts)
        self.likelihood.params = self.posterior.rvs()

    def meanfield_update(self, data, weights=None, sample=True):
        self._set(self._counts(data, weights))
        if sample:
            self.likelihood.params = self.posterior.rvs()

    def meanfield_sgd(self, data, weights, scale, step_size, sample=True):
        counts = self._counts(data, weights)
        acc = stick_acc_counts(counts)
        self.posterior.gammas = (1. - step_size) * self.posterior.gammas\
            + step_size * (self.prior.gammas + 1. / scale * counts)
        self.posterior.deltas = (1. - step_size) * self.posterior.deltas\
            + step_size * (self.prior.deltas + 1. / scale * acc)
        if sample:
            self.likelihood.params = self.posterior.rvs()

    def expected_log_gating(self):
        """E_log_stick_k + sum_{j<k} E_log_rest_j (gmm.py:250-252)."""
        log_stick, log_rest = self.posterior.expected_statistics()
        return log_stick + np.hstack((0, np.cumsum(log_rest)[:-1]))


class _ConjugateBlock:
    """Shared update logic: stats -> posterior natural parameters -> refreshed likelihood."""

    def _apply(self, stats):
        self.posterior.nat_param = self.prior.nat_param + stats

    def _apply_sgd(self, stats, scale, step_size):
        self.posterior.nat_param = (1. - step_size) * self.posterior.nat_param\
            + step_size * (self.prior.nat_param + 1. / scale * stats)

    def variational_lowerbound(self):
        native = getattr(self.posterior, 'native_vlb', None)
        v = native(self.prior) if native is not None else None
        return v if v is not None else self.posterior.entropy() - self.posterior.cross_entropy(self.prior)

    def log_marginal_likelihood(self):
        return self.posterior.log_partition() - self.prior.log_partition()

    def refresh_likelihood(self):
        """likelihood.params = posterior.rvs() — the tail of meanfield_update(sample=True) (bayesian.py:225-230,
        864-869 of the reference) as a separate step, see _CategoricalWrapper.refresh_likelihood."""
        self.likelihood.params = self.posterior.rvs()


class StackedGaussiansWithNormalWisharts(_ConjugateBlock):
    """reference: bayesian.py:182-265 + 268-323 (stacked)."""

    def __init__(self, size, dim, prior, likelihood=None, engine=None):
        self.size = size
        self.dim = dim
        self.prior = prior
        self.posterior = copy.deepcopy(prior)
        if likelihood is None:
            mus, lmbdas = prior.rvs()
            likelihood = StackedGaussiansWithPrecision(size=size, dim=dim, mus=mus, lmbdas=lmbdas, engine=engine)
        self.likelihood = likelihood

    def _stats(self, data, weights):
        return self.likelihood.statistics(data) if weights is None\
            else self.likelihood.weighted_statistics(data, weights)

    def max_aposteriori(self, data, weights=None, stats=None):
        self._apply(stats if stats is not None else self._stats(data, weights))
        self.likelihood.params = self.posterior.mode()

    def resample(self, data, labels=None, stats=None, rng=None):
        self._apply(stats if stats is not None else self._stats(data, labels))
        self.likelihood.params = self.posterior.rvs(rng) if rng is not None else self.posterior.rvs()
        drawn = getattr(self.posterior, 'drawn_canonical', None)
        if drawn is not None and hasattr(self.likelihood, 'adopt_canonical'):
            self.likelihood.adopt_canonical(*drawn)     # (no-op unless the likelihood holds exactly this draw)

    def meanfield_update(self, data, weights=None, stats=None, sample=True):
        """bayesian.py:225-230.  `sample=False` skips the (numerically irrelevant for VI) refresh
        likelihood.params = posterior.rvs() and with it the K Wishart draws it costs per iteration."""
        self._apply(stats if stats is not None else self._stats(data, weights))
        if sample:
            self.likelihood.params = self.posterior.rvs()

    def meanfield_sgd(self, data, weights, scale, step_size, stats=None, sample=True):
        self._apply_sgd(stats if stats is not None else self._stats(data, weights), scale, step_size)
        if sample:
            self.likelihood.params = self.posterior.rvs()

    def canonical_expected(self):
        return self.posterior.canonical_expected()

    def expected_log_likelihood(self, x):
        """(K, N) table <E_q[eta_k], t(x_n)> (bayesian.py:287-301), on the engine."""
        eng = self.likelihood._bind(x)
        eng.estep(*self.canonical_expected(), stats=False, keep_logp=True)
        return eng.get_logp(self.size)

    # ---- posterior predictive (bayesian.py:303-323) ---------------------------------------------
    def posterior_predictive_gaussian(self):
        """Moment-matched Gaussian N(m_k, (df_k/(1 + 1/kappa_k) psi_k)^-1), df = nu - D + 1."""
        mus, kappas, psis, nus = self.posterior.params
        dfs = nus - self.dim + 1
        cs = 1. + 1. / kappas
        return mus, (dfs / cs)[:, None, None] * psis

    def predictive_canonical(self):
        """(c, b, W) of log N(x; posterior_predictive_gaussian()) — stacked_mvn_logpdf (stats.py:53-66)."""
        mus, lmbdas = self.posterior_predictive_gaussian()
        b = np.einsum('kdl,kl->kd', lmbdas, mus)
        c = - 0.5 * np.einsum('kd,kd->k', mus, b) - 0.5 * self.dim * np.log(2. * np.pi)\
            + 0.5 * np.linalg.slogdet(lmbdas)[1]
        return c, b, lmbdas

    def log_posterior_predictive_gaussian(self, x):
        eng = self.likelihood._bind(np.reshape(x, (-1, self.dim)))
        eng.estep(*self.predictive_canonical(), stats=False, keep_logp=True)
        return eng.get_logp(self.size)

    def posterior_predictive_studentt(self):
        mus, lmbdas = self.posterior_predictive_gaussian()
        return mus, lmbdas, self.posterior.nus - self.dim + 1

    def log_posterior_predictive_studentt(self, x):
        raise NotImplementedError(_STUDENTT_DEFECT)


class TiedGaussiansWithNormalWisharts(StackedGaussiansWithNormalWisharts):
    """Gaussians with one shared precision under a TiedNormalWisharts prior/posterior (bayesian.py:326-340).
    The kernels see K identical W_k; the pooling happens in the posterior's nat_to_std."""

    def __init__(self, size, dim, prior, likelihood=None, engine=None):
        if likelihood is None:
            mus, lmbdas = prior.rvs()
            likelihood = TiedGaussiansWithPrecision(size=size, dim=dim, mus=mus, lmbdas=lmbdas, engine=engine)
        super().__init__(size, dim, prior, likelihood, engine=engine)


class StackedGaussiansWithNormalGammas(StackedGaussiansWithNormalWisharts):
    """Diagonal-precision Gaussians under Normal-Gamma priors / posteriors (bayesian.py:343-481).
    Same update skeleton as the Normal-Wishart block; the engine evaluates the same canonical form with a
    diagonal W and the statistic block is reduced to [sum r x, n_d, n_d, sum r x^2] (K, D) on the way in.
    The posterior keeps the reference's observable update rule by default — see StackedNormalGammas."""

    def __init__(self, size, dim, prior, likelihood=None, engine=None):
        self.size = size
        self.dim = dim
        self.prior = prior
        self.posterior = copy.deepcopy(prior)
        if likelihood is None:
            mus, lmbdas_diags = prior.rvs()
            likelihood = self._likelihood_class(size=size, dim=dim, mus=mus, lmbdas_diags=lmbdas_diags,
                                                engine=engine)
        self.likelihood = likelihood

    _likelihood_class = StackedGaussiansWithDiagonalPrecision

    def posterior_predictive_gaussian(self):
        """bayesian.py:457-461: N(m_k, diag((alpha/beta) / (1 + 1/kappa))^-1)."""
        mus, kappas, alphas, betas = self.posterior.params
        return mus, (alphas / betas) / (1. + 1. / kappas)

    def predictive_canonical(self):
        mus, lmbda_diags = self.posterior_predictive_gaussian()
        b = lmbda_diags * mus
        c = - 0.5 * np.sum(mus * b, axis=1) - 0.5 * self.dim * np.log(2. * np.pi)\
            + 0.5 * np.sum(np.log(lmbda_diags), axis=1)
        return c, b, np.eye(self.dim) * lmbda_diags[:, None, :]

    def posterior_predictive_studentt(self):
        mus, lmbda_diags = self.posterior_predictive_gaussian()
        return mus, lmbda_diags, 2. * self.posterior.alphas


class TiedGaussiansWithNormalGammas(StackedGaussiansWithNormalGammas):
    """One diagonal precision shared by all K under a TiedNormalGammas prior / posterior (bayesian.py:484-500)."""

    _likelihood_class = TiedGaussiansWithDiagonalPrecision


class StackedLinearGaussiansWithMatrixNormalWisharts(_ConjugateBlock):
    """reference: bayesian.py:796-912 + 915-985 (stacked)."""

    def __init__(self, size, column_dim, row_dim, prior, likelihood=None, affine=True, engine=None):
        self.size = size
        self.prior = prior
        self.posterior = copy.deepcopy(prior)
        if likelihood is None:
            As, lmbdas = prior.rvs()
            likelihood = StackedLinearGaussiansWithPrecision(size, column_dim, row_dim, As=As, lmbdas=lmbdas,
                                                             affine=affine, engine=engine)
        self.likelihood = likelihood

    def _stats(self, x, y, weights):
        return self.likelihood.statistics(x, y) if weights is None\
            else self.likelihood.weighted_statistics(x, y, weights)

    def max_aposteriori(self, x, y, weights=None, stats=None):
        self._apply(stats if stats is not None else self._stats(x, y, weights))
        self.likelihood.params = self.posterior.mode()

    def resample(self, x, y, z=None, stats=None, rng=None):
        self._apply(stats if stats is not None else self._stats(x, y, z))
        self.likelihood.params = self.posterior.rvs(rng) if rng is not None else self.posterior.rvs()

    def meanfield_update(self, x, y, weights=None, stats=None, sample=True):
        self._apply(stats if stats is not None else self._stats(x, y, weights))
        if sample:
            self.likelihood.params = self.posterior.rvs()

    def meanfield_sgd(self, x, y, weights, scale, step_size, stats=None, sample=True):
        self._apply_sgd(stats if stats is not None else self._stats(x, y, weights), scale, step_size)
        if sample:
            self.likelihood.params = self.posterior.rvs()

    def canonical_expected(self):
        return self.posterior.canonical_expected(affine=self.likelihood.affine)

    def expected_log_likelihood(self, x, y):
        """(K, N) table of the expected log-density of y | x (bayesian.py:933-947), on the engine."""
        eng = self.likelihood._bind(x, y)
        eng.estep(*self.canonical_expected(), stats=False, keep_logp=True)
        return eng.get_logp(self.size)

    # ---- posterior predictive (bayesian.py:949-985) ---------------------------------------------
    def predictive_blocks(self):
        """Per-component blocks of the Gaussian posterior predictive of y | x:
        mean M_k x~, precision P_k / cs_kn with P_k = df_k psi_k, cs_kn = 1 + x~' K_k^-1 x~.
        Returns M, Q = K^-1, Cc = P^-1, P, logdet P."""
        Ms, Ks, psis, nus = self.posterior.params
        dfs = nus - self.likelihood.row_dim + 1
        P = dfs[:, None, None] * psis
        return Ms, np.linalg.inv(Ks), np.linalg.inv(P), P, np.linalg.slogdet(P)[1]

    def _scale_table(self, x):
        """cs (K, N) = 1 + x~' K_k^-1 x~ on the engine (a quadratic form of x: c + b.x - x'Wx/2)."""
        _, Q, _, _, _ = self.predictive_blocks()
        dx = self.likelihood.input_dim
        if self.likelihood.affine:
            c, b, W = 1. + Q[:, dx, dx], 2. * Q[:, :dx, dx], - 2. * Q[:, :dx, :dx]
        else:
            c, b, W = np.ones(self.size), np.zeros((self.size, dx)), - 2. * Q
        from mimo_amd import engine as _engine
        eng = _engine.bind(self.likelihood.engine, np.ascontiguousarray(x))
        eng.estep(c, np.ascontiguousarray(b), np.ascontiguousarray(W), stats=False, keep_logp=True)
        return eng.get_logp(self.size)

    def posterior_predictive_gaussian(self, x):
        """-> mus (K, N, dy), lmbdas (K, N, dy, dy)  (bayesian.py:949-962)."""
        x = np.reshape(x, (-1, self.likelihood.input_dim))
        Ms, _, _, P, _ = self.predictive_blocks()
        xt = np.hstack((x, np.ones((len(x), 1)))) if self.likelihood.affine else x
        mus = np.einsum('kdl,nl->knd', Ms, xt)
        lmbdas = P[:, None, :, :] / self._scale_table(x)[:, :, None, None]
        return mus, lmbdas

    def log_posterior_predictive_gaussian(self, x, y):
        """(K, N) table log N(y_n; M_k x~_n, lmbda_kn^-1): what bayesian.py:964-966 intends (the reference
        passes the (K,N,dy) means to a routine written for (K,dy) means and raises, stats.py:57)."""
        mus, lmbdas = self.posterior_predictive_gaussian(x)
        r = np.reshape(y, (-1, self.likelihood.output_dim))[None, :, :] - mus
        d = r.shape[-1]
        return - 0.5 * np.einsum('knd,kndl,knl->kn', r, lmbdas, r) - 0.5 * d * np.log(2. * np.pi)\
            + 0.5 * np.linalg.slogdet(lmbdas)[1]

    def posterior_predictive_studentt(self, x):
        mus, lmbdas = self.posterior_predictive_gaussian(x)
        return mus, lmbdas, self.posterior.nus - self.likelihood.row_dim + 1

    def log_posterior_predictive_studentt(self, x, y):
        raise NotImplementedError(_STUDENTT_DEFECT)


class TiedLinearGaussiansWithMatrixNormalWisharts(StackedLinearGaussiansWithMatrixNormalWisharts):
    """Linear-Gaussian experts with one shared output precision under a TiedMatrixNormalWisharts
    prior/posterior (bayesian.py:988-1003) — the `models` block of examples/ilr/evaluate_*.py."""

    def __init__(self, size, column_dim, row_dim, prior, likelihood=None, affine=True, engine=None):
        if likelihood is None:
            As, lmbdas = prior.rvs()
            likelihood = TiedLinearGaussiansWithPrecision(size, column_dim, row_dim, As=As, lmbdas=lmbdas,
                                                          affine=affine, engine=engine)
        super().__init__(size, column_dim, row_dim, prior, likelihood, affine=affine, engine=engine)
