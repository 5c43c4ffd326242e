"""Hierarchical Normal-Wishart components: K Gaussians that share ONE precision matrix drawn from a
Normal-Wishart hyper-prior, each mean under a scaled-precision Gaussian prior.

    NormalWishart                                   <-> mimo/distributions/composite.py:19-134
    TiedGaussiansWithScaledPrecision                <-> mimo/distributions/gaussian.py:1038-1202 (+ :890-1035 per block)
    TiedGaussiansWithHierarchicalNormalWisharts     <-> mimo/distributions/bayesian.py:592-793

The reference re-computes `likelihood.weighted_statistics(data, weights)` in every one of its `nb_iter`
fixed-point sub-iterations although neither the data nor the weights change inside that loop
(bayesian.py:633-634, 666-668, 699-700); here the block arrives ONCE from the fused pass over the data
(`stats=`) and the sub-iterations are pure O(K D^3) host algebra.  The expected log-density is the same
canonical form  c_k + b_k.x - 1/2 x'W x  with one shared W = nu psi, so the E-step runs on the same kernels.
"""
import copy

import ctypes as C

import numpy as np
import numpy.random as npr
import scipy.linalg as sla

from mimo_amd.utils.abstraction import Statistics as Stats
from mimo_amd.distributions.wishart import Wishart, wishart_expected_logdet
from mimo_amd.distributions.gaussian import TiedGaussiansWithPrecision


class _Mean:
    """Holder for the Gaussian factor's mean, so that `hyper_prior.gaussian.mu` reads as in the reference."""

    def __init__(self, mu):
        self.mu = mu


class NormalWishart:
    """One Normal-Wishart block over (mu, Lambda): Lambda ~ W(psi, nu), mu | Lambda ~ N(m, (kappa Lambda)^-1)."""

    def __init__(self, dim, mu=None, kappa=None, psi=None, nu=None):
        self.dim = dim
        self.gaussian = _Mean(None if mu is None else np.array(mu, dtype=float))
        self.wishart = Wishart(dim=dim, psi=None if psi is None else np.array(psi, dtype=float), nu=nu)
        self.kappa = kappa

    @property
    def params(self):
        return self.gaussian.mu, self.kappa, self.wishart.psi, self.wishart.nu

    @params.setter
    def params(self, values):
        self.gaussian.mu, self.kappa, self.wishart.psi, self.wishart.nu = values

    @property
    def nat_param(self):
        return self.std_to_nat(self.params)

    @nat_param.setter
    def nat_param(self, natparam):
        self.params = self.nat_to_std(natparam)

    def std_to_nat(self, params):
        """composite.py:50-65."""
        mu, kappa, psi, nu = params
        return Stats([kappa * mu, kappa, np.linalg.inv(psi) + kappa * np.outer(mu, mu), nu - self.dim])

    def nat_to_std(self, natparam):
        """composite.py:67-72."""
        mu = natparam[0] / natparam[1]
        kappa = natparam[1]
        return mu, kappa, np.linalg.inv(natparam[2] - kappa * np.outer(mu, mu)), natparam[3] + self.dim

    def mean(self):
        return self.gaussian.mu, self.wishart.mean()

    def mode(self):
        return self.gaussian.mu, (self.wishart.nu - self.dim) * self.wishart.psi

    def rvs(self):
        """composite.py:82-86 in the reference's RNG order: Wishart (Bartlett), then the mean."""
        lmbda = self.wishart.rvs()
        chol_inv = sla.inv(sla.cholesky(self.kappa * lmbda, lower=False))
        return self.gaussian.mu + npr.normal(size=self.dim).dot(chol_inv.T), lmbda

    def log_base(self):
        return - 0.5 * self.dim * np.log(2. * np.pi)

    def log_partition(self):
        return - 0.5 * self.dim * np.log(self.kappa) + self.wishart.log_partition()

    def expected_statistics(self):
        """composite.py:106-118."""
        nupsi = self.wishart.nu * self.wishart.psi
        E_lmbda_mu = nupsi @ self.gaussian.mu
        return (E_lmbda_mu, - 0.5 * (self.dim / self.kappa + self.gaussian.mu.dot(E_lmbda_mu)), - 0.5 * nupsi,
                0.5 * wishart_expected_logdet(self.wishart.psi, self.wishart.nu))

    @staticmethod
    def _inner(nat, stats):
        return np.dot(nat[0], stats[0]) + nat[1] * stats[1] + np.tensordot(nat[2], stats[2]) + nat[3] * stats[3]

    def entropy(self):
        return self.log_partition() - self.log_base() - self._inner(self.nat_param, self.expected_statistics())

    def cross_entropy(self, dist):
        return dist.log_partition() - dist.log_base() - self._inner(dist.nat_param, self.expected_statistics())


class TiedGaussiansWithScaledPrecision:
    """K Gaussians over the component means: N(mu_k, (kappa_k Lambda_k)^-1), the precision being the
    likelihood's up to the factor kappa_k (gaussian.py:1038-1202).  Batched (K, D) / (K,) / (K, D, D) arrays."""

    def __init__(self, size, dim, kappas, mus=None, lmbdas=None):
        self.size = size
        self.dim = dim
        f = lambda v: None if v is None else np.array(v, dtype=float)
        self.kappas, self.mus = f(kappas), f(mus)
        self._lmbdas, self._chol, self._lmbdas_inv = f(lmbdas), None, None

    @property
    def lmbdas(self):
        return self._lmbdas

    @lmbdas.setter
    def lmbdas(self, value):
        """Assigning the precisions is the ONLY event that drops the cached Cholesky factors of
        omega = kappa * Lambda (gaussian.py:949-958); a later change of kappa alone leaves them in place, and the
        reference's draws and entropies then use the factors of the OLD kappa (bayesian.py:662-664 updates
        posterior.kappas every sweep, :764 reads the entropy).  Kept, because the bound is compared bit for bit."""
        self._lmbdas = None if value is None else np.asarray(value, dtype=float)
        self._chol = self._lmbdas_inv = None

    @property
    def params(self):
        return self.mus, self.kappas

    @params.setter
    def params(self, values):
        self.mus, self.kappas = (np.asarray(v, dtype=float) for v in values)

    @property
    def nat_param(self):
        return self.std_to_nat(self.params)

    @nat_param.setter
    def nat_param(self, natparam):
        self.params = self.nat_to_std(natparam)

    @staticmethod
    def std_to_nat(params):
        mus, kappas = params
        return Stats([kappas[:, None] * mus, kappas])

    @staticmethod
    def nat_to_std(natparam):
        return natparam[0] / natparam[1][:, None], natparam[1]

    @property
    def omegas(self):
        return self.kappas[:, None, None] * self.lmbdas

    @property
    def omegas_chol(self):
        """Upper factors (scipy convention), cached until `lmbdas` is assigned again."""
        if self._chol is None:
            self._chol = np.swapaxes(np.linalg.cholesky(self.omegas), -1, -2)
        return self._chol

    @property
    def sigmas(self):
        return np.linalg.inv(self.omegas)

    def trace_with_sigmas(self, a):
        """tr(a Omega_k^-1) for all k, a symmetric: the inverses of the precisions are kept until `lmbdas` is assigned
        again (the mean-field sweeps change kappa only), so that an iteration costs one (K, D^2) x (D^2,) product."""
        if self._lmbdas_inv is None:
            self._lmbdas_inv = np.linalg.inv(self._lmbdas).reshape(self.size, -1)
        return (self._lmbdas_inv @ np.ravel(a)) / self.kappas

    def rvs(self, sizes):
        """Per block, same RNG use as gaussian.py:975-977."""
        out = []
        chol = self.omegas_chol
        for k, size in enumerate(sizes):
            shape = self.dim if size == 1 else (size, self.dim)
            out.append(self.mus[k] + npr.normal(size=shape).dot(sla.inv(chol[k]).T))
        return np.vstack(out)

    def mean(self):
        return self.mus

    def mode(self):
        return self.mus

    def entropies(self):
        """gaussian.py:1028-1030 per block."""
        half_logdet = np.sum(np.log(np.diagonal(self.omegas_chol, axis1=1, axis2=2)), axis=1)
        return 0.5 * self.dim * np.log(2. * np.pi * np.e) - half_logdet


class TiedGaussiansWithHierarchicalNormalWisharts:
    """reference: bayesian.py:592-793."""

    def __init__(self, size, dim, hyper_prior, prior, engine=None):
        self.size = size
        self.dim = dim
        taus, lmbdas = self._draw_hyper(hyper_prior)
        prior.mus, prior.lmbdas = taus, lmbdas
        self.hyper_prior = hyper_prior
        self.hyper_posterior = copy.deepcopy(hyper_prior)
        self.prior = prior
        self.posterior = copy.deepcopy(prior)
        mus = self.prior.rvs(sizes=self.size * [1])
        self.likelihood = TiedGaussiansWithPrecision(size=size, dim=dim, mus=mus, lmbdas=lmbdas, engine=engine)

    def _draw_hyper(self, dist):
        taus, lmbdas = np.zeros((self.size, self.dim)), np.zeros((self.size, self.dim, self.dim))
        for k in range(self.size):
            taus[k], lmbdas[k] = dist.rvs()
        return taus, lmbdas

    def _stats(self, data, weights, stats):
        return stats if stats is not None else self.likelihood.weighted_statistics(data, weights)

    def _hyper_params(self, mus, xk, nk, xxTk, pooled=None):
        """The pooled hyper-posterior of (tau, Lambda) given the component means `mus`
        (bayesian.py:641-653, identical at :670-682 and :708-720).  `pooled` = `_pooled(xxTk)`, the part that
        does not depend on `mus`, computed once per update by the callers that iterate."""
        hp, K = self.hyper_prior, self.size
        kap, m0 = self.prior.kappas, hp.gaussian.mu
        rho = np.sum(kap[:, None] * mus + hp.kappa * m0, axis=0) / np.sum(kap + hp.kappa)
        kappa = np.sum(kap + hp.kappa) / K
        d = m0[None, :] - mus
        shrink = (d * (hp.kappa * kap / (hp.kappa + kap))[:, None]).T @ d
        cross = mus.T @ xk
        psi = np.linalg.inv((self._pooled(xxTk) if pooled is None else pooled)
                            + (shrink - cross - cross.T + (mus * nk[:, None]).T @ mus) / K)
        nu = np.sum(hp.wishart.nu + nk + 1) / K
        return rho, kappa, psi, nu

    def _pooled(self, xxTk):
        return np.linalg.inv(self.hyper_prior.wishart.psi) + np.sum(xxTk, axis=0) / self.size

    # ---- Gibbs sampling (bayesian.py:619-659) ------------------------------------------------------
    def resample(self, data, labels=None, nb_iter=5, stats=None):
        xk, nk, xxTk, _ = self._stats(data, labels, stats)
        mus = lmbdas = None
        for _ in range(nb_iter):
            taus, lmbdas = self._draw_hyper(self.hyper_posterior)
            self.prior.mus, self.prior.lmbdas = taus, lmbdas
            self.posterior.nat_param = self.prior.nat_param + Stats([xk, nk])
            self.posterior.lmbdas = lmbdas
            mus = self.posterior.rvs(sizes=self.size * [1])
            self.hyper_posterior.params = self._hyper_params(mus, xk, nk, xxTk)
        self.likelihood.mus, self.likelihood.lmbdas = mus, lmbdas

    # ---- mean field (bayesian.py:661-689) -----------------------------------------------------------
    def meanfield_update(self, data, weights=None, nb_iter=25, stats=None):
        xk, nk, xxTk, _ = self._stats(data, weights, stats)
        if not self._meanfield_native(xk, nk, xxTk, nb_iter):
            kap = self.prior.kappas
            pooled = self._pooled(xxTk)
            for _ in range(nb_iter):
                self.posterior.kappas = kap + nk
                self.posterior.mus = (kap[:, None] * self.hyper_posterior.gaussian.mu[None, :] + xk) / (kap + nk)[:, None]
                self.hyper_posterior.params = self._hyper_params(self.posterior.mus, xk, nk, xxTk, pooled)
        _, lmbda = self.hyper_posterior.mode()
        self.likelihood.mus = self.posterior.mode()
        self.likelihood.lmbdas = np.stack(self.size * [lmbda])

    def _meanfield_native(self, xk, nk, xxTk, nb_iter):
        """mimo_host_hier_vi: the nb_iter rounds in one native call (the loop above costs ~0.1 ms of NumPy per round)."""
        from mimo_amd.distributions.composite import _native, _c64, _p
        lib = _native()
        hp, hq, K, D = self.hyper_prior, self.hyper_posterior, self.size, self.dim
        if lib is None or nb_iter < 1 or type(hp) is not NormalWishart or type(hq) is not NormalWishart:
            return False
        kap, m0, xk, nk = _c64(self.prior.kappas), _c64(hp.gaussian.mu), _c64(xk), _c64(nk)
        sxx = _c64(np.sum(xxTk, axis=0))
        psi0_inv = _c64(np.linalg.inv(hp.wishart.psi))
        mu_q = np.array(hq.gaussian.mu, dtype=np.float64)
        if kap.shape != (K,) or m0.shape != (D,) or xk.shape != (K, D) or nk.shape != (K,) or sxx.shape != (D, D) or mu_q.shape != (D,):
            return False
        mus, kappas, psi = np.empty((K, D)), np.empty(K), np.empty((D, D))
        kq, nq = C.c_double(), C.c_double()
        if lib.mimo_host_hier_vi(K, D, int(nb_iter), _p(kap), _p(m0), float(hp.kappa), _p(psi0_inv), float(hp.wishart.nu), _p(xk),
                                 _p(nk), _p(sxx), _p(mu_q), _p(mus), _p(kappas), C.addressof(kq), _p(psi), C.addressof(nq)) != 0:
            return False
        self.posterior.kappas, self.posterior.mus = kappas, mus
        hq.params = (mu_q, kq.value, psi, nq.value)
        return True

    # ---- stochastic mean field (bayesian.py:691-732) --------------------------------------------------
    def meanfield_sgd(self, data, weights, nb_iter, scale, step_size, stats=None):
        xk, nk, xxTk, _ = 1. / scale * Stats(self._stats(data, weights, stats))
        pooled = self._pooled(xxTk)
        for _ in range(nb_iter):
            tau, lmbda = self.hyper_posterior.mean()
            self.prior.mus = np.stack(self.size * [tau])
            self.prior.lmbdas = np.stack(self.size * [lmbda])
            self.posterior.nat_param = (1. - step_size) * self.posterior.nat_param\
                + step_size * (self.prior.nat_param + Stats([xk, nk]))
            self.posterior.lmbdas = np.stack(self.size * [lmbda])
            params = self._hyper_params(self.posterior.mean(), xk, nk, xxTk, pooled)
            self.hyper_posterior.nat_param = (1. - step_size) * self.hyper_posterior.nat_param\
                + step_size * self.hyper_posterior.std_to_nat(params)
        _, lmbda = self.hyper_posterior.mode()
        self.likelihood.mus = self.posterior.mode()
        self.likelihood.lmbdas = np.stack(self.size * [lmbda])

    # ---- E-step form (bayesian.py:734-755) ------------------------------------------------------------
    def canonical_expected(self):
        """(c, b, W): E[Lambda mu_k].x - 1/2 x' E[Lambda] x - 1/2 m_k'E[Lambda]m_k - 1/2 tr(E[Lambda] Omega_k^-1)
        + 1/2 E[logdet Lambda] - D/2 log 2pi, with E[Lambda] = nu psi of the hyper-posterior shared by all k."""
        w = self.hyper_posterior.wishart
        nupsi = w.nu * w.psi
        b = self.posterior.mus @ nupsi.T
        c = - 0.5 * self.dim * np.log(2. * np.pi) - 0.5 * np.einsum('kd,kd->k', self.posterior.mus, b)\
            - 0.5 * self.posterior.trace_with_sigmas(nupsi) + 0.5 * wishart_expected_logdet(w.psi, w.nu)
        return c, b, np.broadcast_to(nupsi, (self.size,) + nupsi.shape)

    def expected_log_likelihood(self, x):
        eng = self.likelihood._bind(x)
        eng.estep(*self.canonical_expected(), stats=False, keep_logp=True)
        return eng.get_logp(self.size)

    def variational_lowerbound(self):
        """bayesian.py:757-783 (a scalar: the sum over the K blocks)."""
        hq, w = self.hyper_posterior, self.hyper_posterior.wishart
        nupsi = w.nu * w.psi
        kap = self.prior.kappas
        d = self.posterior.mus - hq.gaussian.mu[None, :]
        per_k = self.posterior.entropies() - 0.5 * self.dim * np.log(2. * np.pi) + 0.5 * self.dim * np.log(kap)\
            + 0.5 * wishart_expected_logdet(w.psi, w.nu) - 0.5 * kap * self.dim / hq.kappa\
            - 0.5 * kap * np.einsum('kd,kd->k', d @ nupsi, d)\
            - 0.5 * kap * self.posterior.trace_with_sigmas(nupsi)
        return self.size * (hq.entropy() - hq.cross_entropy(self.hyper_prior)) + np.sum(per_k)

    # ---- posterior predictive (bayesian.py:785-793) ---------------------------------------------------
    def posterior_predictive_gaussian(self):
        w = self.hyper_posterior.wishart
        return self.posterior.mus, np.stack(self.size * [(w.nu - self.dim + 1) * w.psi])

    def predictive_canonical(self):
        mus, lmbdas = self.posterior_predictive_gaussian()
        b = np.einsum('kdl,kl->kd', lmbdas, mus)
        c = - 0.5 * np.einsum('kd,kd->k', mus, b) - 0.5 * self.dim * np.log(2. * np.pi)\
            + 0.5 * np.linalg.slogdet(lmbdas)[1]
        return c, b, lmbdas

    def log_posterior_predictive_gaussian(self, x):
        eng = self.likelihood._bind(np.reshape(x, (-1, self.dim)))
        eng.estep(*self.predictive_canonical(), stats=False, keep_logp=True)
        return eng.get_logp(self.size)


# ---------------------------------------------------------------------------------------------
# Hierarchical linear-Gaussian experts: one slope matrix and one output precision shared by all K
# experts, a separate offset per expert (mimo/distributions/bayesian.py:1222-1522)
# ---------------------------------------------------------------------------------------------
class MatrixNormalWithPrecision:
    """vec(A) ~ N(vec(M), (K (x) V)^-1), A of shape (row_dim, column_dim) (matrix.py:10-175)."""

    def __init__(self, column_dim, row_dim, M=None, V=None, K=None):
        self.column_dim = column_dim
        self.row_dim = row_dim
        f = lambda v: None if v is None else np.array(v, dtype=float)
        self.M, self.V, self.K = f(M), f(V), f(K)

    @property
    def params(self):
        return self.M, self.V, self.K

    @params.setter
    def params(self, values):
        self.M, self.V, self.K = values

    def mean(self):
        return self.M

    def mode(self):
        return self.M

    def rvs(self):
        """matrix.py:122-124: one normal(row_dim * column_dim) call, Fortran-order reshape."""
        chol_inv = sla.inv(sla.cholesky(np.kron(self.K, self.V), lower=False))
        aux = npr.normal(size=self.row_dim * self.column_dim).dot(chol_inv.T)
        return self.M + np.reshape(aux, (self.row_dim, self.column_dim), order='F')


class StackedAffineLinearGaussiansWithPrecision:
    """K experts y | x ~ N(A_k x + c_k, Lambda_k^-1) with slope and offset kept apart
    (lingauss.py:576-744).  The engine sees the joint row z = [x, y] and the same quadratic form as the
    affine experts of lingauss.py with A~ = [A | c]."""

    affine = True

    def __init__(self, size, column_dim, row_dim, As=None, cs=None, lmbdas=None, engine=None):
        self.size = size
        self.column_dim = column_dim
        self.row_dim = row_dim
        f = lambda v: None if v is None else np.array(v, dtype=float)
        self.As, self.cs, self.lmbdas = f(As), f(cs), f(lmbdas)
        self._engine = engine

    @property
    def engine(self):
        from mimo_amd import engine as _engine
        return self._engine if self._engine is not None else _engine.default_engine()

    @property
    def params(self):
        return self.As, self.cs, self.lmbdas

    @params.setter
    def params(self, values):
        self.As, self.cs, self.lmbdas = (np.asarray(v, dtype=float) for v in values)

    @property
    def input_dim(self):
        return self.column_dim

    @property
    def output_dim(self):
        return self.row_dim

    @property
    def lmbdas_chol(self):
        return np.swapaxes(np.linalg.cholesky(self.lmbdas), -1, -2)

    @property
    def lmbdas_chol_inv(self):
        return np.stack([sla.inv(c) for c in self.lmbdas_chol])

    def predict(self, x):
        """lingauss.py:647-649."""
        return np.einsum('kdl,...l->k...d', self.As, x) + self.cs[:, None, :]

    def log_base(self):
        return - 0.5 * self.row_dim * np.log(2. * np.pi) * np.ones(self.size)

    def canonical(self):
        from mimo_amd.distributions.lingauss import residual_quadratic
        aug = np.concatenate([self.As, self.cs[:, :, None]], axis=2)
        c0, b, W = residual_quadratic(aug, self.lmbdas, True, self.column_dim)
        logdet_half = np.sum(np.log(np.diagonal(self.lmbdas_chol, axis1=1, axis2=2)), axis=1)
        return c0 + logdet_half + self.log_base(), b, W

    def _bind(self, x, y):
        from mimo_amd import engine as _engine
        from mimo_amd.distributions.lingauss import joint_rows
        x = np.asarray(x, dtype=float).reshape(-1, self.column_dim)
        y = np.asarray(y, dtype=float).reshape(-1, self.row_dim)
        return _engine.bind(self.engine, joint_rows(x, y))

    def log_likelihood(self, x, y):
        eng = self._bind(x, y)
        eng.estep(*self.canonical(), stats=False, keep_logp=True)
        return eng.get_logp(self.size)

    @staticmethod
    def block_stats(S, dx):
        """engine block over z = [x, y] -> Stats([ymk, xmk, yxTk, xxTk, yyTk, nk]) (lingauss.py:700-715)."""
        return Stats([S.sx[:, dx:], S.sx[:, :dx], S.sxx[:, dx:, :dx], S.sxx[:, :dx, :dx], S.sxx[:, dx:, dx:], S.n])

    def weighted_statistics(self, x, y, weights):
        eng = self._bind(x, y)
        return self.block_stats(eng.weighted_stats(np.asarray(weights, dtype=float)), self.column_dim)


class TiedAffineLinearGaussiansWithMatrixNormalWisharts:
    """reference: bayesian.py:1222-1522.  The reference re-contracts the data in every sub-iteration
    (nine einsums over N, :1263-1273, :1325-1335); here the joint statistics block of the fused pass
    (`stats=`) is read once and the sub-iterations are O(K d^3) host algebra: with c_k the current offsets,
    sum r c c' = n_k c c',  sum r y c' = ymk c',  sum r c x' = c xmk',
    sum r (y - c)(y - c)' = yyTk - ymk c' - c ymk' + n_k c c'."""

    def __init__(self, size, column_dim, row_dim, slope_prior, offset_prior, precision_prior, likelihood=None,
                 engine=None, reference_rng=True):
        self.size = size
        self.column_dim = column_dim
        self.row_dim = row_dim
        self.reference_rng = reference_rng
        As = np.zeros((size, row_dim, column_dim))
        lmbdas = np.zeros((size, row_dim, row_dim))
        for k in range(size):
            lmbdas[k] = precision_prior.rvs()
            slope_prior.V = lmbdas[k]
            As[k] = slope_prior.rvs()
        offset_prior.lmbdas = lmbdas
        cs = offset_prior.rvs(sizes=size * [1])
        self.slope_prior, self.offset_prior, self.precision_prior = slope_prior, offset_prior, precision_prior
        self.likelihood = likelihood if likelihood is not None else\
            StackedAffineLinearGaussiansWithPrecision(size, column_dim, row_dim, As, cs, lmbdas, engine=engine)
        self.slope_posterior = copy.deepcopy(slope_prior)
        self.offset_posterior = copy.deepcopy(offset_prior)
        self.precision_posterior = copy.deepcopy(precision_prior)

    def _stats(self, x, y, weights, stats):
        return stats if stats is not None else self.likelihood.weighted_statistics(x, y, weights)

    def _slope_and_precision(self, cs, ymk, xmk, yxTk, xxTk, yyTk, nk):
        """Pooled posterior of the shared slope (M, K) and precision (psi, nu) given the offsets `cs`
        (bayesian.py:1275-1298, identical at :1337-1360)."""
        Kn = self.size
        M0, K0 = self.slope_prior.M, self.slope_prior.K
        psi0, nu0 = self.precision_prior.psi, self.precision_prior.nu
        cxTk = np.einsum('kd,kl->kdl', cs, xmk)
        B = (M0 @ K0)[None, :, :] + yxTk - cxTk
        Kk_inv = np.linalg.inv(K0[None, :, :] + xxTk)
        M = np.sum(B @ Kk_inv, axis=0) / Kn
        K = np.sum(K0[None, :, :] + xxTk, axis=0) / Kn
        resid = yyTk - np.einsum('kd,kl->kdl', ymk, cs) - np.einsum('kd,kl->kdl', cs, ymk)\
            + nk[:, None, None] * np.einsum('kd,kl->kdl', cs, cs)
        dc = cs - self.offset_prior.mus
        psi = np.linalg.inv(np.linalg.inv(psi0) + M0 @ K @ M0.T + np.sum(resid, axis=0) / Kn
                            + np.sum(self.offset_prior.kappas[:, None, None] * np.einsum('kd,kl->kdl', dc, dc), axis=0) / Kn
                            - np.sum(B @ Kk_inv @ np.swapaxes(B, 1, 2), axis=0) / Kn)
        nu = np.sum(nu0 + nk + 1) / Kn
        return M, K, psi, nu

    def _offsets(self, A_k, ymk, xmk, nk):
        kap = self.offset_prior.kappas
        rhos = (kap[:, None] * self.offset_prior.mus + (ymk - np.einsum('kdl,kl->kd', A_k, xmk))) / (kap + nk)[:, None]
        return rhos, kap + nk

    # ---- Gibbs sampling (bayesian.py:1258-1318) ------------------------------------------------------
    def resample(self, x, y, z=None, nb_iter=25, stats=None):
        ymk, xmk, yxTk, xxTk, yyTk, nk = self._stats(x, y, z, stats)
        As = cs = lmbdas = None
        for _ in range(nb_iter):
            cs = self.offset_posterior.rvs(sizes=self.size * [1])
            M, K, psi, nu = self._slope_and_precision(cs, ymk, xmk, yxTk, xxTk, yyTk, nk)
            self.slope_posterior.M, self.slope_posterior.K = M, K
            self.precision_posterior.psi, self.precision_posterior.nu = psi, nu
            As = np.zeros((self.size, self.row_dim, self.column_dim))
            lmbdas = np.zeros((self.size, self.row_dim, self.row_dim))
            for k in range(self.size):
                lmbdas[k] = self.precision_posterior.rvs()
                self.slope_posterior.V = lmbdas[k]
                As[k] = self.slope_posterior.rvs()
            self.offset_posterior.mus, self.offset_posterior.kappas = self._offsets(As, ymk, xmk, nk)
            self.offset_posterior.lmbdas = lmbdas
        self.likelihood.As, self.likelihood.cs, self.likelihood.lmbdas = As, cs, lmbdas

    # ---- mean field (bayesian.py:1320-1385) -------------------------------------------------------------
    def meanfield_update(self, x, y, weights=None, nb_iter=25, stats=None):
        ymk, xmk, yxTk, xxTk, yyTk, nk = self._stats(x, y, weights, stats)
        for _ in range(nb_iter):
            cs = self.offset_posterior.mean()
            M, K, psi, nu = self._slope_and_precision(cs, ymk, xmk, yxTk, xxTk, yyTk, nk)
            self.slope_posterior.M, self.slope_posterior.K = M, K
            self.precision_posterior.psi, self.precision_posterior.nu = psi, nu
            lmbda = self.precision_posterior.mean()
            self.slope_posterior.V = lmbda
            A = self.slope_posterior.mean()
            self.offset_posterior.mus, self.offset_posterior.kappas =\
                self._offsets(np.broadcast_to(A, (self.size,) + A.shape), ymk, xmk, nk)
            self.offset_posterior.lmbdas = np.stack(self.size * [lmbda])
        A, lmbda = self.slope_posterior.mode(), self.precision_posterior.mode()
        self.likelihood.As = np.stack(self.size * [A])
        self.likelihood.lmbdas = np.stack(self.size * [lmbda])
        self.likelihood.cs = self.offset_posterior.mode()

    def meanfield_sgd(self, x, y, weights, nb_iter, scale, step_size, stats=None):
        raise NotImplementedError        # bayesian.py:1387-1388

    # ---- the equivalent stacked Matrix-Normal-Wishart blocks (bayesian.py:1394-1419, :1453-1476) ----------
    def _as_mnw(self, slope, offset, precision):
        from mimo_amd.distributions.composite import StackedMatrixNormalWisharts
        Kn, dx, dy = self.size, self.column_dim, self.row_dim
        Ms = np.concatenate([np.broadcast_to(slope.M, (Kn, dy, dx)), offset.mus[:, :, None]], axis=2)
        Ks = np.zeros((Kn, dx + 1, dx + 1))
        Ks[:, :dx, :dx] = slope.K
        Ks[:, dx, dx] = offset.kappas
        return StackedMatrixNormalWisharts(Kn, dx + 1, dy, Ms=Ms, Ks=Ks, psis=np.stack(Kn * [precision.psi]),
                                           nus=np.stack(Kn * [precision.nu]).astype(float))

    def prior_mnw(self):
        return self._as_mnw(self.slope_prior, self.offset_prior, self.precision_prior)

    def posterior_mnw(self):
        return self._as_mnw(self.slope_posterior, self.offset_posterior, self.precision_posterior)

    def reference_draw(self):
        """The reference builds a throw-away StackedLinearGaussiansWithMatrixNormalWisharts for every table of
        expected log-densities and every predictive call (bayesian.py:1410-1413, :1503-1506); that constructor
        initialises a likelihood from `prior.rvs()` (bayesian.py:922-926): K Wishart and K matrix-normal draws
        from the global stream per E-step.  Drawn and discarded here (`reference_rng=True`, the default) so that
        whatever a seeded run draws next — the random initialisation of the next inner mixture, a later Gibbs
        sweep — sees the reference's stream."""
        if self.reference_rng:
            self.prior_mnw().rvs()

    def canonical_expected(self):
        """One call = one `expected_log_likelihood(x, y)` of the reference (see reference_draw)."""
        self.reference_draw()
        return self.posterior_mnw().canonical_expected(affine=True)

    def expected_log_likelihood(self, x, y):
        eng = self.likelihood._bind(x, y)
        eng.estep(*self.canonical_expected(), stats=False, keep_logp=True)
        return eng.get_logp(self.size)

    def variational_lowerbound(self):
        post, prior = self.posterior_mnw(), self.prior_mnw()
        return post.entropy() - post.cross_entropy(prior)

    # ---- posterior predictive (bayesian.py:1478-1522, through the same equivalent blocks) -----------------
    def predictive_blocks(self):
        """M~_k = [M | mu_k], Q = K~_k^-1, Cc = P^-1, P = df psi, logdet P — the inputs of mimo_predict."""
        self.reference_draw()
        q = self.posterior_mnw()
        P = (q.nus - self.row_dim + 1)[:, None, None] * q.psis
        return q.Ms, np.linalg.inv(q.Ks), np.linalg.inv(P), P, np.linalg.slogdet(P)[1]

    def posterior_predictive_gaussian(self, x):
        """-> mus (K, N, dy), lmbdas (K, N, dy, dy) (bayesian.py:949-962 on the equivalent blocks)."""
        x = np.reshape(x, (-1, self.column_dim))
        Ms, Q, _, P, _ = self.predictive_blocks()
        xt = np.hstack((x, np.ones((len(x), 1))))
        cs = 1. + np.einsum('nd,kdl,nl->kn', xt, Q, xt)
        return np.einsum('kdl,nl->knd', Ms, xt), P[:, None, :, :] / cs[:, :, None, None]
