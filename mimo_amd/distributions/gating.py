"""Gating distributions: Categorical likelihood, Dirichlet and truncated stick-breaking priors.

Same class / method names and semantics as the reference's mimo/distributions/categorical.py and
mimo/distributions/dirichlet.py; everything here is O(K) host math (float64 NumPy / SciPy).  The
only O(N) quantity the gating needs — the per-component counts sum_n r_kn / bincount(labels) — is
one column of the sufficient-statistic block the HIP engine returns.
"""
import warnings

import numpy as np
import numpy.random as npr
from scipy.special import digamma, gammaln, betaln


class Categorical:
    """reference: mimo/distributions/categorical.py:5-68"""

    def __init__(self, dim, probs=None):
        self.dim = dim
        self.probs = probs
        if probs is None:
            self.probs = 1. / self.dim * np.ones((self.dim,))

    @property
    def params(self):
        return self.probs

    @params.setter
    def params(self, values):
        self.probs = values

    @property
    def nb_params(self):
        return len(self.probs) - 1

    def mode(self):
        return np.argmax(self.probs)

    def rvs(self, size=1):
        return npr.choice(a=self.dim, p=self.probs, size=size)          # categorical.py:32

    def statistics(self, data):
        """Label counts, one per class (categorical.py:35-39).  A list of label arrays is pooled."""
        if isinstance(data, np.ndarray):
            return np.bincount(data, minlength=self.dim)
        total = np.zeros(self.dim, dtype=np.int64)
        for part in data:
            total = total + self.statistics(part)
        return total

    def weighted_statistics(self, data, weights):
        """sum_n r_kn (categorical.py:41-46).  `data` is unused, as in the reference; `weights` is either the
        (K,) vector of counts the engine already reduced (what the drivers pass) or a (K,N) table, or a list
        of such tables whose sums are pooled."""
        if isinstance(weights, np.ndarray):
            return weights if weights.ndim == 1 else np.atleast_2d(weights).sum(axis=1)
        total = np.zeros(self.dim)
        for table in weights:
            total = total + self.weighted_statistics(None, table)
        return total

    def log_likelihood(self, x):
        """log probs[x_n]; NaN labels score 0 (categorical.py:51-59)."""
        x = np.asarray(x)
        known = ~np.isnan(x)
        out = np.zeros(x.shape, dtype=np.double)
        with np.errstate(divide='ignore', invalid='ignore'):
            out[known] = np.log(self.probs)[x[known].astype(np.intp)]
        return out

    def max_likelihood(self, data, weights=None):
        """probs = normalised counts (categorical.py:65-68)."""
        n = self.statistics(data) if weights is None else self.weighted_statistics(data, weights)
        self.probs = n / np.sum(n)


class Dirichlet:
    """reference: mimo/distributions/dirichlet.py:8-97"""

    def __init__(self, dim=None, alphas=None):
        self.dim = dim
        self.alphas = alphas

    @property
    def params(self):
        return self.alphas

    @params.setter
    def params(self, values):
        self.alphas = values

    @property
    def nat_param(self):
        return self.std_to_nat(self.params)

    @nat_param.setter
    def nat_param(self, natparam):
        self.params = self.nat_to_std(natparam)

    @staticmethod
    def std_to_nat(params):
        return params - 1.

    @staticmethod
    def nat_to_std(natparam):
        return natparam + 1.

    def mean(self):
        return self.alphas / np.sum(self.alphas)

    def mode(self):
        assert np.all(self.alphas > 1.), "Make sure alphas > 1."
        return (self.alphas - 1.) / (np.sum(self.alphas) - self.dim)

    def rvs(self, size=1):
        return npr.dirichlet(self.alphas)

    @property
    def base(self):
        return 1.

    def log_base(self):
        return np.log(self.base)

    def log_partition(self):
        return np.add.reduce(gammaln(self.alphas)) - gammaln(np.add.reduce(self.alphas))

    def log_likelihood(self, x):
        return - self.log_partition() + self.log_base() + np.sum((self.alphas - 1.) * np.log(x))

    def expected_statistics(self):
        return digamma(self.alphas) - digamma(np.add.reduce(self.alphas))

    def entropy(self):
        return self.log_partition() - self.log_base() - self.nat_param.dot(self.expected_statistics())

    def cross_entropy(self, dist):
        return dist.log_partition() - dist.log_base() - dist.nat_param.dot(self.expected_statistics())

    def entropy_minus_cross_entropy(self, dist):
        """entropy() - cross_entropy(dist), the expected statistics computed once (same terms, same order)."""
        E = self.expected_statistics()
        return (self.log_partition() - self.log_base() - self.nat_param.dot(E))\
            - (dist.log_partition() - dist.log_base() - dist.nat_param.dot(E))


class TruncatedStickBreaking:
    """reference: mimo/distributions/dirichlet.py:100-214 (Ishwaran & James 2001; Blei & Jordan 2006)"""

    def __init__(self, dim=None, gammas=None, deltas=None):
        self.dim = dim
        self.gammas = gammas
        self.deltas = deltas

    @property
    def params(self):
        return self.gammas, self.deltas

    @params.setter
    def params(self, values):
        self.gammas, self.deltas = values

    @property
    def nat_param(self):
        return self.std_to_nat(self.params)

    @nat_param.setter
    def nat_param(self, natparam):
        self.params = self.nat_to_std(natparam)

    @staticmethod
    def std_to_nat(params):
        return params[0] - 1., params[1] - 1.

    @staticmethod
    def nat_to_std(natparam):
        return natparam[0] + 1., natparam[1] + 1.

    @staticmethod
    def _probs(betas_head):
        betas = np.hstack((betas_head, 1.))
        probs = np.zeros((betas.shape[0],))
        probs[0] = betas[0]
        probs[1:] = betas[1:] * np.cumprod(1.0 - betas[:-1])
        return probs

    def mean(self):
        return self._probs(self.gammas[:-1] / (self.gammas[:-1] + self.deltas[:-1]))

    def mode(self):
        g, d = self.gammas[:-1], self.deltas[:-1]
        betas = np.full(g.shape, np.nan)
        both = (g > 1.) & (d > 1.)
        betas[both] = (g[both] - 1.) / (g[both] + d[both] - 2.)
        betas[(g == 1.) & (d == 1.)] = 1.
        betas[(g < 1.) & (d < 1.)] = 1.
        betas[(g <= 1.) & (d > 1.)] = 0.
        betas[(g > 1.) & (d <= 1.)] = 1.
        if np.isnan(betas).any():
            warnings.warn("Mode of Dirichlet process not defined")
            raise ValueError
        return self._probs(betas)

    def rvs(self, size=1, truncate=True):
        return self._probs(npr.beta(self.gammas[:-1], self.deltas[:-1]))      # dirichlet.py:177-186

    @property
    def base(self):
        return 1.

    def log_base(self):
        return np.log(self.base)

    def log_partition(self):
        return np.sum(betaln(self.gammas, self.deltas))

    def expected_statistics(self):
        E_log_stick = digamma(self.gammas) - digamma(self.gammas + self.deltas)
        E_log_rest = digamma(self.deltas) - digamma(self.gammas + self.deltas)
        return E_log_stick, E_log_rest

    def entropy(self):
        nat, stats = self.nat_param, self.expected_statistics()
        return self.log_partition() - self.log_base() - (nat[0].dot(stats[0]) + nat[1].dot(stats[1]))

    def cross_entropy(self, dist):
        nat, stats = dist.nat_param, self.expected_statistics()
        return dist.log_partition() - dist.log_base() - (nat[0].dot(stats[0]) + nat[1].dot(stats[1]))
