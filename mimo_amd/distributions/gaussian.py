"""Stacked Gaussian components with full precision matrices.

API mirror of the reference's StackedGaussiansWithPrecision (mimo/distributions/gaussian.py:377-542),
re-designed around batched (K, D, D) arrays: host code keeps the O(K D^3) parameter math in float64
NumPy/LAPACK; every O(N) method hands (c, b, W) of the canonical form

    l[k,n] = c_k + b_k.x_n - 1/2 x_n' W_k x_n          (W = Lambda, b = Lambda mu,
    c = -(1/2 mu'Lambda mu - sum log diag chol Lambda) - D/2 log 2pi   for point estimates)

to the HIP engine (mimo_amd.engine.HipEngine) and gets the (K, N) table / packed statistics back.
"""
import numpy as np
import numpy.random as npr
import scipy.linalg as sla

from mimo_amd.utils.abstraction import Statistics as Stats
from mimo_amd import engine as _engine


def symmetrize(A):
    return (A + np.swapaxes(A, -1, -2)) / 2.


def upper_chol(mats):
    """Upper Cholesky factors U (A = U'U) of a stack, like scipy.linalg.cholesky(lower=False)."""
    return np.swapaxes(np.linalg.cholesky(mats), -1, -2)


class StackedGaussiansWithPrecision:

    diagonal = False
    tied = False

    @staticmethod
    def block_stats(S):
        """engine block -> Stats([sum r x, n, sum r xx', n]) (gaussian.py:502)."""
        return Stats([S.sx, S.n, S.sxx, S.n])

    def __init__(self, size, dim, mus=None, lmbdas=None, engine=None):
        self.size = size
        self.dim = dim
        self.mus = None if mus is None else np.array(mus, dtype=float)
        self.lmbdas = None if lmbdas is None else np.array(lmbdas, dtype=float)
        self._engine = engine

    # ---- engine -------------------------------------------------------------------------------
    @property
    def engine(self):
        return self._engine if self._engine is not None else _engine.default_engine()

    @engine.setter
    def engine(self, value):
        self._engine = value

    # ---- parameters ---------------------------------------------------------------------------
    @property
    def params(self):
        return self.mus, self.lmbdas

    @params.setter
    def params(self, values):
        self.mus, self.lmbdas = (np.asarray(v, dtype=float) for v in values)

    @property
    def nat_param(self):
        return self.std_to_nat(self.params)

    @nat_param.setter
    def nat_param(self, natparam):
        self.params = self.nat_to_std(natparam)

    def std_to_nat(self, params):
        mus, lmbdas = params
        return Stats([np.einsum('kdl,kl->kd', lmbdas, mus), - 0.5 * lmbdas])

    def nat_to_std(self, natparam):
        mus = - 0.5 * np.einsum('kdl,kl->kd', np.linalg.inv(natparam[1]), natparam[0])
        return mus, - 2. * natparam[1]

    @property
    def lmbdas_chol(self):
        return upper_chol(self.lmbdas)

    @property
    def lmbdas_chol_inv(self):
        return np.stack([sla.inv(c) for c in self.lmbdas_chol])

    @property
    def sigmas(self):
        ci = self.lmbdas_chol_inv
        return ci @ np.swapaxes(ci, -1, -2)

    def mean(self):
        return self.mus

    def mode(self):
        return self.mus

    def rvs(self, sizes):
        """Per-component draws, same RNG use as gaussian.py:311-313 / :448-449."""
        ci = self.lmbdas_chol_inv
        out = []
        for k, size in enumerate(sizes):
            shape = self.dim if size == 1 else (size, self.dim)
            out.append(self.mus[k] + npr.normal(size=shape).dot(ci[k].T))
        return np.vstack(out)

    @property
    def base(self):
        return np.power(2. * np.pi, - self.dim / 2.) * np.ones(self.size)

    def log_base(self):
        return np.log(self.base)

    def log_partition(self):
        """gaussian.py:352-354 per component: 1/2 mu'Lambda mu - sum log diag chol(Lambda)."""
        quad = np.einsum('kd,kdl,kl->k', self.mus, self.lmbdas, self.mus)
        return 0.5 * quad - np.sum(np.log(np.diagonal(self.lmbdas_chol, axis1=1, axis2=2)), axis=1)

    # ---- canonical form for the engine --------------------------------------------------------
    def canonical(self):
        """(c, b, W) of the point-estimate log-density (Gibbs / EM form, gaussian.py:510-521)."""
        memo = getattr(self, '_canon', None)
        if memo is not None and memo[0] is self.mus and memo[1] is self.lmbdas:
            return memo[2].copy(), memo[3], self.lmbdas
        b = np.einsum('kdl,kl->kd', self.lmbdas, self.mus)
        return - self.log_partition() + self.log_base(), b, self.lmbdas

    def adopt_canonical(self, mus, lmbdas, c, b):
        """(c, b) computed by whoever produced (mus, lmbdas) — the native Gibbs draw (mimo_host_nw_gibbs); kept only
        while the parameters are these very arrays, so any later assignment falls back to the formulas above."""
        self._canon = (mus, lmbdas, c, b) if (mus is self.mus and lmbdas is self.lmbdas) else None

    # ---- O(N) methods: on the engine ----------------------------------------------------------
    @property
    def structure(self):
        """Feature map the engine runs for this family (mimo_set_structure)."""
        return 'diag' if self.diagonal else 'linear' if self.tied else 'full'

    def _bind(self, data):
        # rows with NaN: the library drops them from the statistics and gives them the normaliser-only log-density,
        # as the reference does (gaussian.py:493-494, 512-520; include/mimo_hip.h, mimo_nan_info)
        data = np.asarray(data, dtype=float)
        return _engine.bind(self.engine, data.reshape(-1, self.dim), self.structure)

    def log_likelihood(self, x):
        """(K, N) table of component log-densities (gaussian.py:510-521)."""
        if not isinstance(x, np.ndarray):
            return list(map(self.log_likelihood, x))
        eng = self._bind(x)
        eng.estep(*self.canonical(), stats=False, keep_logp=True)
        return eng.get_logp(self.size)

    def weighted_statistics(self, data, weights):
        """Stats([sum r x, n, sum r xx', n]) (gaussian.py:491-502); n appears twice because it feeds
        both kappa and nu."""
        if not isinstance(data, np.ndarray):
            stats = list(map(self.weighted_statistics, data, weights))
            out = stats[0]
            for s in stats[1:]:
                out = out + s
            return out
        eng = self._bind(data)
        return self.block_stats(eng.weighted_stats(np.asarray(weights, dtype=float)))

    def statistics(self, data, fold=True):
        """gaussian.py:466-489.  fold=True: the data totals replicated for every component.
        fold=False would materialise (K, N, D, D) (13 GB at N=1e4, D=16, K=64); the engine evaluates
        <E[eta_k], t(x_n)> without it, so that form is not offered."""
        if not fold:
            raise NotImplementedError("statistics(fold=False) materialises (K,N,D,D); use "
                                      "expected_log_likelihood / the fused E-step instead")
        eng = self._bind(data)
        S = eng.weighted_stats(np.ones((1, eng.N)))
        rep = lambda a: np.repeat(a, self.size, axis=0)
        return Stats([rep(S.sx), rep(S.n), rep(S.sxx), rep(S.n)])

    # ---- M-step -------------------------------------------------------------------------------
    def max_likelihood(self, data, weights=None, stats=None):
        """gaussian.py:525-542; `stats` lets the drivers pass engine output directly."""
        xk, nk, xxTk, _ = stats if stats is not None else self.weighted_statistics(data, weights)
        mus = xk / nk[:, None]
        sigmas = xxTk / nk[:, None, None] - np.einsum('kd,kl->kdl', mus, mus)
        sigmas = symmetrize(sigmas) + 1e-16 * np.eye(self.dim)
        assert np.all(np.linalg.eigvalsh(sigmas) > 0.)
        self.mus, self.lmbdas = mus, np.linalg.inv(sigmas)


class TiedGaussiansWithPrecision(StackedGaussiansWithPrecision):
    """K Gaussians sharing one precision matrix (gaussian.py:545-572): pooled covariance in the M-step.
    With all W_k equal the quadratic term leaves the softmax, and every update of a tied block only uses
    sum_k of the second-moment blocks — a constant of the data (or of the row weights).  The engine therefore
    runs the Dz + 1 feature kernels ('linear' structure) and hands over that pooled matrix."""

    tied = True

    @staticmethod
    def block_stats(S):
        if S.sxx is None:       # 'linear' structure: K equal shares of the pooled second moment (only their sum is used)
            K = S.n.shape[0]
            return Stats([S.sx, S.n, np.broadcast_to(S.sxx_total / K, (K,) + S.sxx_total.shape), S.n])
        return Stats([S.sx, S.n, S.sxx, S.n])

    def max_likelihood(self, data, weights=None, stats=None):
        xk, nk, xxTk, _ = stats if stats is not None else self.weighted_statistics(data, weights)
        mus = xk / nk[:, None]
        sigma = (np.sum(xxTk, axis=0) - np.einsum('k,kd,kl->dl', nk, mus, mus)) / np.sum(nk)
        sigma = symmetrize(sigma) + 1e-16 * np.eye(self.dim)
        assert np.all(np.linalg.eigvalsh(sigma) > 0.)
        self.mus, self.lmbdas = mus, np.array(self.size * [np.linalg.inv(sigma)])


class StackedGaussiansWithDiagonalPrecision(StackedGaussiansWithPrecision):
    """K Gaussians with diagonal precisions, parameters (mus (K,D), lmbdas_diags (K,D))
    (mimo/distributions/gaussian.py:697-852).  The engine sees the same canonical form with
    W_k = diag(lambda_k); the statistic block it returns is reduced to its diagonal here."""

    diagonal = True

    def __init__(self, size, dim, mus=None, lmbdas_diags=None, engine=None):
        self.size = size
        self.dim = dim
        self.mus = None if mus is None else np.array(mus, dtype=float)
        self.lmbdas_diags = None if lmbdas_diags is None else np.array(lmbdas_diags, dtype=float)
        self._engine = engine

    @property
    def params(self):
        return self.mus, self.lmbdas_diags

    @params.setter
    def params(self, values):
        self.mus, self.lmbdas_diags = (np.asarray(v, dtype=float) for v in values)

    def std_to_nat(self, params):
        mus, lmbdas_diags = params
        return Stats([lmbdas_diags * mus, - 0.5 * lmbdas_diags])

    def nat_to_std(self, natparam):
        return - 0.5 * natparam[0] / natparam[1], - 2. * natparam[1]

    @property
    def lmbdas(self):
        return np.eye(self.dim) * self.lmbdas_diags[:, None, :]

    @property
    def lmbdas_chol(self):
        return np.eye(self.dim) * np.sqrt(self.lmbdas_diags)[:, None, :]

    @property
    def lmbdas_chol_inv(self):
        return np.eye(self.dim) / np.sqrt(self.lmbdas_diags)[:, None, :]

    @property
    def sigmas_diags(self):
        return 1. / self.lmbdas_diags

    @property
    def sigmas(self):
        return np.eye(self.dim) * self.sigmas_diags[:, None, :]

    def log_partition(self):
        """gaussian.py:678-680 per component."""
        return 0.5 * np.sum(self.lmbdas_diags * self.mus**2, axis=1) - 0.5 * np.sum(np.log(self.lmbdas_diags), axis=1)

    def canonical(self):
        return - self.log_partition() + self.log_base(), self.lmbdas_diags * self.mus, self.lmbdas

    @staticmethod
    def block_stats(S):
        """engine block -> Stats([sum r x, n_d, n_d, sum r x^2]) with (K, D) entries (gaussian.py:802-815)."""
        nd = np.repeat(S.n[:, None], S.sx.shape[1], axis=1)
        return Stats([S.sx, nd, nd, np.diagonal(S.sxx, axis1=1, axis2=2).copy()])

    def weighted_statistics(self, data, weights):
        if not isinstance(data, np.ndarray):
            stats = list(map(self.weighted_statistics, data, weights))
            out = stats[0]
            for s in stats[1:]:
                out = out + s
            return out
        eng = self._bind(data)
        return self.block_stats(eng.weighted_stats(np.asarray(weights, dtype=float)))

    def statistics(self, data, fold=True):
        """gaussian.py:777-800, fold=True (see StackedGaussiansWithPrecision.statistics for fold=False)."""
        if not fold:
            raise NotImplementedError("statistics(fold=False) materialises (K,N,D); use "
                                      "expected_log_likelihood / the fused E-step instead")
        eng = self._bind(data)
        x, nd, _, xx = self.block_stats(eng.weighted_stats(np.ones((1, eng.N))))
        rep = lambda a: np.repeat(a, self.size, axis=0)
        return Stats([rep(x), rep(nd), rep(nd), rep(xx)])

    def max_likelihood(self, data, weights=None, stats=None):
        """gaussian.py:841-852."""
        xk, ndk, _, xxk = stats if stats is not None else self.weighted_statistics(data, weights)
        mus = xk / ndk
        self.mus, self.lmbdas_diags = mus, 1. / (xxk / ndk - mus**2 + 1e-16)


class TiedGaussiansWithDiagonalPrecision(StackedGaussiansWithDiagonalPrecision):
    """One diagonal precision shared by all K (gaussian.py:855-878): pooled variance in the M-step."""

    def max_likelihood(self, data, weights=None, stats=None):
        xk, ndk, _, xxk = stats if stats is not None else self.weighted_statistics(data, weights)
        mus = xk / ndk
        sigma_diag = (np.sum(xxk, axis=0) - np.sum(ndk * mus**2, axis=0)) / np.sum(ndk, axis=0)
        self.mus = mus
        self.lmbdas_diags = np.array(self.size * [1. / (sigma_diag + 1e-16)])
