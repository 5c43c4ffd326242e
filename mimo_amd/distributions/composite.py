"""Conjugate priors / posteriors of the component blocks, batched over K (float64, host).

StackedNormalWisharts        <-> mimo/distributions/composite.py:19-256
StackedMatrixNormalWisharts  <-> mimo/distributions/composite.py:550-783 (+ matrix.py:10-175)
TiedNormalWisharts           <-> mimo/distributions/composite.py:259-283   (one precision shared by all K)
TiedMatrixNormalWisharts     <-> mimo/distributions/composite.py:786-808

These classes produce the per-component parameter block the kernels consume (expected statistics
-> canonical (c, b, W)) and consume the sufficient-statistic block the kernels produce
(nat_param = prior.nat_param + stats).  K small D x D problems per sweep: they stay on the host.
"""
import numpy as np
import numpy.random as npr
import scipy.linalg as sla
from scipy.special import gammaln, digamma

from mimo_amd.utils.abstraction import Statistics as Stats
from mimo_amd.distributions.wishart import (bartlett_variates_in_reference_order, wishart_from_bartlett, legacy_draws,
                                            wishart_log_partition, wishart_expected_logdet, wishart_rvs,
                                            wishart_rvs_batched, sum_log_diag_chol)


def _outer(a, b):
    return np.einsum('kd,kl->kdl', a, b)


# ---------------------------------------------------------------------------------------------
# Native batched route for the per-sweep posterior update (mimo_amd/csrc/mimo_host.cpp): natural
# parameters -> standard parameters, expected statistics and the canonical (c, b, W) in ONE call
# instead of ~40 NumPy calls on the critical path between two kernel launches.  Same formulas as
# the NumPy methods below (which stay the reference-order implementation and the route taken when
# a block is not positive definite, so errors surface exactly as before).  MIMO_HOST_NATIVE=0
# switches it off.
# ---------------------------------------------------------------------------------------------
import ctypes as _ctypes
import os as _os

NATIVE_HOST = _os.environ.get("MIMO_HOST_NATIVE", "1") != "0"


def _native():
    if not NATIVE_HOST:
        return None
    try:
        from mimo_amd import _lib
        return _lib.load()
    except Exception:
        return None


def _p(a, _from_buffer=_ctypes.c_char.from_buffer, _addressof=_ctypes.addressof):
    """plain address of an array's first element: 0.4 us through the buffer protocol (ctypes' data_as() costs ~7 us per
    array, __array_interface__ 1.2 us); read-only or empty arrays take the slower way."""
    try:
        return _addressof(_from_buffer(a))
    except (TypeError, ValueError, BufferError):
        return a.__array_interface__['data'][0]


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class StackedNormalWisharts:
    """K independent Normal-Wishart distributions over (mu_k, Lambda_k):
    Lambda ~ W(psi, nu), mu | Lambda ~ N(m, (kappa Lambda)^-1)."""

    def __init__(self, size, dim, mus=None, kappas=None, psis=None, nus=None):
        self.size = size
        self.dim = dim
        self.mus = None if mus is None else np.array(mus, dtype=float)
        self.kappas = None if kappas is None else np.array(kappas, dtype=float)
        self.psis = None if psis is None else np.array(psis, dtype=float)
        self.nus = None if nus is None else np.array(nus, dtype=float)
        self._memo = {}

    @property
    def params(self):
        return self.mus, self.kappas, self.psis, self.nus

    @params.setter
    def params(self, values):
        self.mus, self.kappas, self.psis, self.nus = (np.asarray(v, dtype=float) for v in values)

    def _cached(self, name, fn):
        """Derived quantities (natural parameters, expectations, log-partition) are recomputed only
        when a parameter ARRAY is replaced; in-place edits of a parameter array are not tracked."""
        key = tuple(map(id, self.params))      # (this runs ~13 times per iteration; shared by the Matrix-Normal-Wishart class)
        memo = self.__dict__.get('_memo')
        if memo is None:
            memo = self._memo = {}
        if memo.get('key') != key:
            memo.clear()
            memo['key'] = key
            memo['refs'] = self.params        # keep the arrays alive so ids cannot be recycled
        try:
            return memo[name]
        except KeyError:
            v = memo[name] = fn()
            return v

    def _set_memo(self, **values):
        """the derived quantities of the parameters just assigned, in one go (what a sequence of _cached(name, lambda: value)
        calls would store; each of those recomputes the key of the four parameter arrays)."""
        params = self.params
        key = tuple(map(id, params))
        memo = self.__dict__.get('_memo')
        if memo is None or memo.get('key') != key:
            memo = self._memo = {'key': key, 'refs': params}
        memo.update(values)

    @property
    def nat_param(self):
        return self._cached('nat', lambda: self.std_to_nat(self.params))

    @nat_param.setter
    def nat_param(self, natparam):
        if not self._assign_native(natparam):
            self.params = self.nat_to_std(natparam)
        self._set_memo(nat=Stats(natparam))   # the exact natural parameters just assigned

    def _assign_native(self, natparam):
        """mimo_host_nw_vi: standard parameters + every derived quantity the sweep needs, in one call."""
        lib = _native()
        if lib is None or type(self).nat_to_std is not StackedNormalWisharts.nat_to_std:
            return False
        a, b, c, d = (_c64(v) for v in natparam)
        K, D = self.size, self.dim
        if a.shape != (K, D) or b.shape != (K,) or c.shape != (K, D, D) or d.shape != (K,):
            return False
        mus, psis, nus, hld = np.empty((K, D)), np.empty((K, D, D)), np.empty(K), np.empty(K)
        cc, bb, W, E2, E4 = np.empty(K), np.empty((K, D)), np.empty((K, D, D)), np.empty(K), np.empty(K)
        if lib.mimo_host_nw_vi(K, D, _p(a), _p(b), _p(c), _p(d), _p(mus), _p(psis), _p(nus), _p(hld),
                               _p(cc), _p(bb), _p(W), _p(E2), _p(E4)) != 0:
            return False
        self.params = (mus, b.copy(), psis, nus)
        self._set_memo(hld=hld, estats=(bb, E2, - 0.5 * W, E4), canon=(cc, bb, W), native=True)
        return True

    def native_vlb(self, prior):
        """entropy() - cross_entropy(prior) per block through mimo_host_nw_vlb, when this posterior was assigned by the
        native sweep (its natural parameters, expectations and half log-determinants are then at hand); None otherwise."""
        memo = self.__dict__.get('_memo') or {}
        if memo.get('key') != tuple(map(id, self.params)) or not memo.get('native'):   # (native: contiguous float64 blocks)
            return None
        lib = _native()
        if lib is None or not isinstance(prior, StackedNormalWisharts) or (prior.size, prior.dim) != (self.size, self.dim):
            return None
        q = [_c64(v) for v in memo['nat']]
        pn = prior._cached('nat_c64', lambda: tuple(_c64(v) for v in prior.nat_param))
        plz = prior._cached('logZ_c64', lambda: _c64(prior.log_partition()))
        E1, E2, _, E4 = memo['estats']
        out = np.empty(self.size)
        if lib.mimo_host_nw_vlb(self.size, self.dim, _p(q[0]), _p(q[1]), _p(q[2]), _p(q[3]), _p(pn[0]), _p(pn[1]),
                                _p(pn[2]), _p(pn[3]), _p(plz), _p(self.nus), _p(memo['hld']), _p(E1), _p(E2),
                                _p(memo['canon'][2]), _p(E4), _p(out)) != 0:
            return None
        return out

    def std_to_nat(self, params):
        """eta = [kappa m, kappa, psi^-1 + kappa m m', nu - D]  (composite.py:50-65)."""
        mus, kappas, psis, nus = params
        return Stats([kappas[:, None] * mus, kappas,
                      np.linalg.inv(psis) + kappas[:, None, None] * _outer(mus, mus),
                      nus - self.dim])

    def nat_to_std(self, natparam):
        """composite.py:67-72."""
        a, b, c, d = natparam
        mus = a / b[:, None]
        psis = np.linalg.inv(c - b[:, None, None] * _outer(mus, mus))
        return mus, b, psis, d + self.dim

    def mean(self):
        return self.mus, self.nus[:, None, None] * self.psis

    def mode(self):
        """composite.py:77-80: Lambda = (nu - D) psi."""
        return self.mus, (self.nus - self.dim)[:, None, None] * self.psis

    def rvs(self, rng=None):
        """Per component, in the reference's RNG order (composite.py:82-86, wishart.py:72-92,
        gaussian.py:311-313): Wishart draw, then mu = m + normal(D) . chol_upper(kappa Lambda)^-T.
        With `rng` (a numpy Generator) all K draws are batched (same law, different stream)."""
        if rng is not None:
            drawn = self._rvs_native(rng)
            if drawn is not None:
                return drawn
            lmbdas = wishart_rvs_batched(self.psis, self.nus, rng)
            # mu = m + L^-T eps with kappa Lambda = L L'  =>  cov = (kappa Lambda)^-1
            L = np.linalg.cholesky(self.kappas[:, None, None] * lmbdas)
            eps = rng.standard_normal((self.size, self.dim, 1))
            return self.mus + np.linalg.solve(np.swapaxes(L, 1, 2), eps)[..., 0], lmbdas
        # numpy.random in the reference's order per component; the O(K D^3) algebra after the draws is batched
        # (it used to be K Python iterations of cholesky + inv: 2.6 ms at K = 64, D = 16 on the critical path of
        # every mean-field iteration of the reference-shaped driver)
        lower, diag, eps = bartlett_variates_in_reference_order(self.nus, self.dim, self.dim)
        drawn = self._draw_algebra_native(lower, diag, eps)       # the same algebra in one native call (mimo_host_nw_gibbs)
        if drawn is not None:
            return drawn
        lmbdas = wishart_from_bartlett(self.psis, lower, diag)
        # mu = m + eps . U^-T with kappa Lambda = U'U (upper factor), i.e. U mu_c = eps
        U = np.swapaxes(np.linalg.cholesky(self.kappas[:, None, None] * lmbdas), 1, 2)
        return self.mus + np.linalg.solve(U, eps[..., None])[..., 0], lmbdas

    def _draw_algebra_native(self, z, g, eps):
        """Lambda_k = T T' with T = chol(psi_k) A_k and mu_k = m_k + (sqrt(kappa_k) T)^-T eps_k from the variates of the reference's
        stream (bartlett_variates_in_reference_order) — mimo_host_nw_gibbs does the three batched factorisations and the solve of
        the NumPy route below in one call (0.3 ms of the 0.7 ms a K = 64, D = 16 draw costs; the variates themselves stay numpy's,
        call by call).  None: library missing, a subclass with its own draw, or a block that is not positive definite."""
        lib = _native()
        if lib is None or type(self).rvs is not StackedNormalWisharts.rvs:
            return None
        K, D = self.size, self.dim
        z, g, eps = _c64(z), _c64(g), _c64(eps)
        mus, kappas, psis = _c64(self.mus), _c64(self.kappas), _c64(self.psis)
        mu, lmbda, c, b = np.empty((K, D)), np.empty((K, D, D)), np.empty(K), np.empty((K, D))
        if lib.mimo_host_nw_gibbs(K, D, _p(mus), _p(kappas), _p(psis), _p(z), _p(g), _p(eps),
                                  _p(mu), _p(lmbda), _p(c), _p(b)) != 0:
            return None
        self.drawn_canonical = (mu, lmbda, c, b)
        return mu, lmbda

    def _rvs_native(self, rng):
        """mimo_host_nw_gibbs: the K Bartlett draws, the K conditional Gaussian draws and the canonical (c, b, W) of
        the drawn Gaussians in one call — three batched Cholesky factorisations, a batched solve and ~20 NumPy calls
        less between two label kernels.  The variates come from `rng` (one normal block, one chi-square block), the
        law is that of the NumPy route below; `self.drawn_canonical` carries (mu, Lambda, c, b) for the likelihood
        that receives the draw (StackedGaussiansWithNormalWisharts.resample)."""
        lib = _native()
        if lib is None or type(self).rvs is not StackedNormalWisharts.rvs:
            return None
        K, D = self.size, self.dim
        nt = D * (D - 1) // 2
        zz = rng.standard_normal((K, nt + D))
        z, eps = np.ascontiguousarray(zz[:, :nt]), np.ascontiguousarray(zz[:, nt:])
        g = np.sqrt(rng.chisquare(np.asarray(self.nus)[:, None] - np.arange(D)[None, :]))
        mus, kappas, psis = _c64(self.mus), _c64(self.kappas), _c64(self.psis)
        mu, lmbda, c, b = np.empty((K, D)), np.empty((K, D, D)), np.empty(K), np.empty((K, D))
        if lib.mimo_host_nw_gibbs(K, D, _p(mus), _p(kappas), _p(psis), _p(z), _p(g), _p(eps),
                                  _p(mu), _p(lmbda), _p(c), _p(b)) != 0:
            return None       # a block that is not positive definite: the NumPy route raises as before
        self.drawn_canonical = (mu, lmbda, c, b)
        return mu, lmbda

    @property
    def base(self):
        return np.power(2. * np.pi, - self.dim / 2.) * np.ones(self.size)

    def log_base(self):
        return np.log(self.base)

    def _half_logdet_psi(self):
        return self._cached('hld', lambda: sum_log_diag_chol(self.psis))

    def log_partition(self):
        """composite.py:95-98."""
        return self._cached('logZ', lambda: - 0.5 * self.dim * np.log(self.kappas)
                            + wishart_log_partition(self.psis, self.nus, self._half_logdet_psi()))

    def expected_statistics(self):
        """E[Lambda mu], E[-1/2 mu'Lambda mu], E[-1/2 Lambda], E[1/2 logdet Lambda] (composite.py:106-118)."""
        return self._cached('estats', self._expected_statistics)

    def _expected_statistics(self):
        nupsi = self.nus[:, None, None] * self.psis
        E_lmbda_mu = np.einsum('kdl,kl->kd', nupsi, self.mus)
        E_muT_lmbda_mu = - 0.5 * (self.dim / self.kappas + np.einsum('kd,kd->k', self.mus, E_lmbda_mu))
        return (E_lmbda_mu, E_muT_lmbda_mu, - 0.5 * nupsi,
                0.5 * wishart_expected_logdet(self.psis, self.nus, self._half_logdet_psi()))

    def canonical_expected(self):
        """(c, b, W) of <E_q[eta_k], t(x)> + log_base  (bayesian.py:287-301): the VI E-step form."""
        def numpy_route():
            E1, E2, E3, E4 = self.expected_statistics()
            return self.log_base() + E2 + E4, E1, - 2. * E3
        return self._cached('canon', numpy_route)

    @staticmethod
    def _inner(nat, stats):
        return (np.einsum('kd,kd->k', nat[0], stats[0]) + nat[1] * stats[1]
                + np.einsum('kdl,kdl->k', nat[2], stats[2]) + nat[3] * stats[3])

    def entropy(self):
        return self.log_partition() - self.log_base() - self._inner(self.nat_param, self.expected_statistics())

    def cross_entropy(self, other):
        return other.log_partition() - other.log_base() - self._inner(other.nat_param, self.expected_statistics())

    def log_likelihood(self, x):
        """sum_k log NW(mu_k, Lambda_k) (composite.py:100-104, 244-246); x = (mus, lmbdas)."""
        mus, lmbdas = x
        D = self.dim
        diff = mus - self.mus
        kl = self.kappas[:, None, None] * lmbdas
        gauss = - 0.5 * np.einsum('kd,kdl,kl->k', diff, kl, diff) + 0.5 * np.linalg.slogdet(kl)[1]\
            - 0.5 * D * np.log(2. * np.pi)
        wish = 0.5 * (self.nus - D - 1) * np.linalg.slogdet(lmbdas)[1]\
            - 0.5 * np.trace(np.linalg.solve(self.psis, lmbdas), axis1=1, axis2=2)\
            - wishart_log_partition(self.psis, self.nus)
        return np.sum(gauss + wish)


class StackedMatrixNormalWisharts:
    """K independent Matrix-Normal-Wishart distributions over (A_k, Lambda_k):
    Lambda ~ W(psi, nu), vec(A) | Lambda ~ N(vec(M), (K (x) Lambda)^-1)."""

    def __init__(self, size, column_dim, row_dim, Ms=None, Ks=None, psis=None, nus=None):
        self.size = size
        self.column_dim = column_dim
        self.row_dim = row_dim
        self.Ms = None if Ms is None else np.array(Ms, dtype=float)
        self.Ks = None if Ks is None else np.array(Ks, dtype=float)
        self.psis = None if psis is None else np.array(psis, dtype=float)
        self.nus = None if nus is None else np.array(nus, dtype=float)
        self._memo = {}

    @property
    def params(self):
        return self.Ms, self.Ks, self.psis, self.nus

    @params.setter
    def params(self, values):
        self.Ms, self.Ks, self.psis, self.nus = (np.asarray(v, dtype=float) for v in values)

    _cached = StackedNormalWisharts._cached
    _set_memo = StackedNormalWisharts._set_memo

    @property
    def nat_param(self):
        return self._cached('nat', lambda: self.std_to_nat(self.params))

    @nat_param.setter
    def nat_param(self, natparam):
        if not self._assign_native(natparam):
            self.params = self.nat_to_std(natparam)
        self._set_memo(nat=Stats(natparam))

    def _assign_native(self, natparam):
        """mimo_host_mnw_vi (affine and non-affine canonical forms are both kept)."""
        lib = _native()
        if lib is None or type(self).nat_to_std is not StackedMatrixNormalWisharts.nat_to_std:
            return False
        a, b, c, d = (_c64(v) for v in natparam)
        K, dy, dc = self.size, self.row_dim, self.column_dim
        if a.shape != (K, dy, dc) or b.shape != (K, dc, dc) or c.shape != (K, dy, dy) or d.shape != (K,) or dc < 2:
            return False
        Ms, psis, nus, hld, Kinv = np.empty((K, dy, dc)), np.empty((K, dy, dy)), np.empty(K), np.empty(K), np.empty((K, dc, dc))
        E1, E2, E4 = np.empty((K, dy, dc)), np.empty((K, dc, dc)), np.empty(K)
        Dz = dc - 1 + dy
        cc, bb, W = np.empty(K), np.empty((K, Dz)), np.empty((K, Dz, Dz))
        if lib.mimo_host_mnw_vi(K, dy, dc, 1, _p(a), _p(b), _p(c), _p(d), _p(Ms), _p(psis), _p(nus), _p(hld), _p(Kinv),
                                _p(cc), _p(bb), _p(W), _p(E1), _p(E2), _p(E4)) != 0:
            return False
        self.params = (Ms, b.copy(), psis, nus)
        self._set_memo(hld=hld, estats=(E1, E2, - 0.5 * nus[:, None, None] * psis, E4), canon_affine=(cc, bb, W))
        return True

    def std_to_nat(self, params):
        """eta = [M K, K, psi^-1 + M K M', nu - d - 1 + l]  (composite.py:577-592)."""
        Ms, Ks, psis, nus = params
        MK = Ms @ Ks
        return Stats([MK, Ks, np.linalg.inv(psis) + MK @ np.swapaxes(Ms, 1, 2),
                      nus - self.row_dim - 1. + self.column_dim])

    def nat_to_std(self, natparam):
        """composite.py:594-599."""
        a, b, c, d = natparam
        Ms = a @ np.linalg.inv(b)
        psis = np.linalg.inv(c - Ms @ b @ np.swapaxes(Ms, 1, 2))
        return Ms, b, psis, d + self.row_dim + 1. - self.column_dim

    def mean(self):
        return self.Ms, self.nus[:, None, None] * self.psis

    def mode(self):
        return self.Ms, (self.nus - self.row_dim)[:, None, None] * self.psis

    def rvs(self, rng=None):
        """Reference RNG order (composite.py:607-611, matrix.py:122-125): Wishart draw, then
        vec_F(A) = vec_F(M) + normal(d l) . chol_upper(kron(K, Lambda))^-T.
        With `rng` all K draws are batched: A = M + Ll^-T E Lk^-1 with Lambda = Ll Ll', K = Lk Lk'."""
        if rng is not None:
            lmbdas = wishart_rvs_batched(self.psis, self.nus, rng)
            Ll, Lk = np.linalg.cholesky(lmbdas), np.linalg.cholesky(self.Ks)
            E = rng.standard_normal((self.size, self.row_dim, self.column_dim))
            X = np.linalg.solve(np.swapaxes(Ll, 1, 2), E)                       # Ll^-T E
            X = np.swapaxes(np.linalg.solve(Lk, np.swapaxes(X, 1, 2)), 1, 2)    # ... Lk^-1
            return self.Ms + X, lmbdas
        # numpy.random in the reference's order per component, algebra batched.  The reference factorises the
        # (dy dx) x (dy dx) matrix kron(K, Lambda) per component; its upper Cholesky factor is kron(Uk, Ul) with
        # K = Uk'Uk, Lambda = Ul'Ul, so  vec_F(aux) = kron(Uk, Ul)^-1 z  is  aux = Ul^-1 Z Uk^-T  with z = vec_F(Z).
        dy, dx = self.row_dim, self.column_dim
        lower, diag, eps = bartlett_variates_in_reference_order(self.nus, dy, dy * dx)
        lmbdas = wishart_from_bartlett(self.psis, lower, diag)
        Z = np.swapaxes(eps.reshape(self.size, dx, dy), 1, 2)                 # order='F' reshape to (dy, dx)
        Ul = np.swapaxes(np.linalg.cholesky(lmbdas), 1, 2)
        Uk = np.swapaxes(np.linalg.cholesky(self.Ks), 1, 2)
        X = np.linalg.solve(Ul, Z)                                              # Ul^-1 Z
        X = np.swapaxes(np.linalg.solve(Uk, np.swapaxes(X, 1, 2)), 1, 2)       # ... Uk^-T
        return self.Ms + X, lmbdas

    @property
    def base(self):
        return np.power(2. * np.pi, - self.row_dim * self.column_dim / 2.) * np.ones(self.size)

    def log_base(self):
        return np.log(self.base)

    def _half_logdet_psi(self):
        return self._cached('hld', lambda: sum_log_diag_chol(self.psis))

    def log_partition(self):
        """composite.py:622-625."""
        return self._cached('logZ', lambda: - 0.5 * self.row_dim * np.linalg.slogdet(self.Ks)[1]
                            + wishart_log_partition(self.psis, self.nus, self._half_logdet_psi()))

    def expected_statistics(self):
        """E[Lambda A], E[-1/2 A'Lambda A], E[-1/2 Lambda], E[1/2 logdet Lambda] (composite.py:635-647)."""
        return self._cached('estats', self._expected_statistics)

    def _expected_statistics(self):
        nupsi = self.nus[:, None, None] * self.psis
        E_Lmbda_A = nupsi @ self.Ms
        E_AT_Lmbda_A = - 0.5 * (self.row_dim * np.linalg.inv(self.Ks) + np.swapaxes(self.Ms, 1, 2) @ E_Lmbda_A)
        return (E_Lmbda_A, E_AT_Lmbda_A, - 0.5 * nupsi,
                0.5 * wishart_expected_logdet(self.psis, self.nus, self._half_logdet_psi()))

    def canonical_expected(self, affine=True):
        """(c, b, W) over z = [x, y] of the expected log-density of y | x (bayesian.py:933-947):
        <E[Lambda A], y x~'> + <E[-1/2 A'Lambda A], x~ x~'> + <E[-1/2 Lambda], y y'> + E[1/2 logdet] + log_base
        with x~ = [x, 1] when affine."""
        if affine:
            hit = self.__dict__.get('_memo', {})
            if hit.get('key') == tuple(id(p) for p in self.params) and 'canon_affine' in hit:
                return hit['canon_affine']
        E1, E2, E3, E4 = self.expected_statistics()
        dy, dc = self.row_dim, self.column_dim
        dx = dc - 1 if affine else dc
        Kc = self.size
        W = np.zeros((Kc, dx + dy, dx + dy))
        b = np.zeros((Kc, dx + dy))
        c = np.power(2. * np.pi, - dy / 2.) * np.ones(Kc)
        c = np.log(c) + E4
        W[:, :dx, :dx] = - 2. * E2[:, :dx, :dx]
        W[:, dx:, dx:] = - 2. * E3
        W[:, dx:, :dx] = - E1[:, :, :dx]            # the cross term <E1, y x'> = -1/2 z'Wz with
        W[:, :dx, dx:] = - np.swapaxes(E1[:, :, :dx], 1, 2)   # W_yx = -E1_x (and its transpose)
        if affine:
            b[:, :dx] = 2. * E2[:, :dx, dx]         # x~ x~' off-diagonal (x, 1) appears twice
            b[:, dx:] = E1[:, :, dx]
            c = c + E2[:, dx, dx]
        return c, b, W

    @staticmethod
    def _inner(nat, stats):
        return (np.einsum('kdl,kdl->k', nat[0], stats[0]) + np.einsum('kdl,kdl->k', nat[1], stats[1])
                + np.einsum('kdl,kdl->k', nat[2], stats[2]) + nat[3] * stats[3])

    def entropy(self):
        return self.log_partition() - self.log_base() - self._inner(self.nat_param, self.expected_statistics())

    def cross_entropy(self, other):
        return other.log_partition() - other.log_base() - self._inner(other.nat_param, self.expected_statistics())


class _TiedNatParam:
    """Tied posteriors pool the Wishart block over k in nat_to_std, so the natural parameters read back
    are those of the pooled standard parameters, not the ones assigned (the reference recomputes them on
    every read: composite.py:166-172) — the assigned block must not be memoised."""

    @property
    def nat_param(self):
        return self._cached('nat', lambda: self.std_to_nat(self.params))

    @nat_param.setter
    def nat_param(self, natparam):
        self.params = self.nat_to_std(natparam)


class TiedNormalWisharts(_TiedNatParam, StackedNormalWisharts):
    """K Normal-Wisharts whose Wishart factor is shared: psi = inv(mean_k(psi_k^-1 block)), nu = mean_k nu_k
    (composite.py:273-283)."""

    def nat_to_std(self, natparam):
        a, b, c, d = natparam
        mus = a / b[:, None]
        psi = np.linalg.inv(np.mean(c - b[:, None, None] * _outer(mus, mus), axis=0))
        nu = np.mean(d + self.dim)
        return mus, b, np.array(self.size * [psi]), np.array(self.size * [nu])

    @property
    def nat_param(self):
        return self._cached('nat', lambda: self.std_to_nat(self.params))

    @nat_param.setter
    def nat_param(self, natparam):
        if not self._assign_native_tied(natparam):
            self.params = self.nat_to_std(natparam)

    def _assign_native_tied(self, natparam):
        """mimo_host_nw_vi_tied: the pooled standard parameters, the natural parameters they read back as, and every
        derived quantity of the sweep in one call."""
        lib = _native()
        if lib is None or type(self).nat_to_std is not TiedNormalWisharts.nat_to_std:
            return False
        a, b, c, d = (_c64(v) for v in natparam)
        K, D = self.size, self.dim
        if a.shape != (K, D) or b.shape != (K,) or c.shape != (K, D, D) or d.shape != (K,):
            return False
        mus, psis, nus, hld, nat_c = np.empty((K, D)), np.empty((K, D, D)), np.empty(K), np.empty(K), np.empty((K, D, D))
        cc, bb, W, E2, E4 = np.empty(K), np.empty((K, D)), np.empty((K, D, D)), np.empty(K), np.empty(K)
        if lib.mimo_host_nw_vi_tied(K, D, _p(a), _p(b), _p(c), _p(d), _p(mus), _p(psis), _p(nus), _p(hld), _p(nat_c),
                                    _p(cc), _p(bb), _p(W), _p(E2), _p(E4)) != 0:
            return False
        kap = b.copy()
        self.params = (mus, kap, psis, nus)
        self._set_memo(nat=Stats([kap[:, None] * mus, kap, nat_c, nus - D]), hld=hld, estats=(bb, E2, - 0.5 * W, E4),
                       canon=(cc, bb, W), native=True)
        return True


class TiedMatrixNormalWisharts(_TiedNatParam, StackedMatrixNormalWisharts):
    """composite.py:798-808."""

    def nat_to_std(self, natparam):
        a, b, c, d = natparam
        Ms = a @ np.linalg.inv(b)
        psi = np.linalg.inv(np.mean(c - Ms @ b @ np.swapaxes(Ms, 1, 2), axis=0))
        nu = np.mean(d + self.row_dim + 1. - self.column_dim)
        return Ms, b, np.array(self.size * [psi]), np.array(self.size * [nu])


# ---------------------------------------------------------------------------------------------
# Diagonal-precision blocks: Normal-Gamma priors / posteriors
# ---------------------------------------------------------------------------------------------
class StackedNormalGammas:
    """K independent Normal-Gamma distributions over (mu_k, diag Lambda_k), every array (K, D):
    lambda_kd ~ Gamma(alpha_kd, beta_kd), mu_kd | lambda_kd ~ N(m_kd, (kappa_kd lambda_kd)^-1)
    (mimo/distributions/composite.py:286-404 per block, :407-519 stacked).

    `reference_setters` (default True) keeps the reference's observable update rule: its stacked
    setters for `alphas` / `betas` write attributes nothing reads (composite.py:472-484), so assigning
    natural or standard parameters moves (mus, kappas) only and the Gamma factors stay where the
    constructor put them — the posterior of every stacked diagonal model keeps the PRIOR's (alpha, beta)
    for the whole run.  Parity is against that behaviour; `reference_setters=False` applies the
    conjugate update to all four parameters."""

    def __init__(self, size, dim, mus=None, kappas=None, alphas=None, betas=None, reference_setters=True):
        self.size = size
        self.dim = dim
        self.reference_setters = reference_setters
        f = lambda v: None if v is None else np.array(v, dtype=float)
        self.mus, self.kappas, self.alphas, self.betas = f(mus), f(kappas), f(alphas), f(betas)

    @property
    def params(self):
        return self.mus, self.kappas, self.alphas, self.betas

    @params.setter
    def params(self, values):
        mus, kappas, alphas, betas = (np.asarray(v, dtype=float) for v in values)
        self.mus, self.kappas = mus, kappas
        if not self.reference_setters:
            self.alphas, self.betas = alphas, betas

    @property
    def nat_param(self):
        return self.std_to_nat(self.params)

    @nat_param.setter
    def nat_param(self, natparam):
        self.params = self.nat_to_std(natparam)

    def std_to_nat(self, params):
        """eta = [kappa m, kappa, 2 alpha - 1, 2 beta + kappa m^2]  (composite.py:314-329)."""
        mus, kappas, alphas, betas = params
        return Stats([kappas * mus, kappas, 2. * alphas - 1., 2. * betas + kappas * mus**2])

    def nat_to_std(self, natparam):
        """composite.py:331-337."""
        a, b, c, d = natparam
        mus = a / b
        return mus, b, 0.5 * (c + 1.), 0.5 * (d - b * mus**2)

    def mean(self):
        return self.mus, self.alphas / self.betas

    def mode(self):
        """composite.py:342-345: lambda = (alpha - 1/2) / beta."""
        return self.mus, (self.alphas - 0.5) / self.betas

    def rvs(self, rng=None):
        """Per block in the reference's RNG order (composite.py:347-351, gamma.py:53-55,
        gaussian.py:646-648): D gamma draws, then mu = m + normal(D) / sqrt(kappa lambda).
        With `rng` (a numpy Generator) all K blocks are drawn at once (same law, different stream)."""
        if rng is not None:
            lmbdas = rng.gamma(self.alphas, 1. / self.betas)
            return self.mus + rng.standard_normal(self.mus.shape) / np.sqrt(self.kappas * lmbdas), lmbdas
        # the generator is called per block as the reference calls it (D gammas, then D normals); the algebra after the loop is
        # the same arithmetic for all K at once: z.dot(diag(s).T) adds exact zeros to z_i s_i
        # gamma(a, scale) IS scale * standard_gamma(a) in the legacy generator: K blocks of (D gammas, D normals) in one call
        _, gam, zs = legacy_draws(0, self.alphas, self.dim)
        lmbdas = (1. / self.betas) * gam
        return self.mus + zs * (1. / np.sqrt(self.kappas * lmbdas)), lmbdas

    @property
    def base(self):
        return np.power(2. * np.pi, - self.dim / 2.) * np.ones(self.size)

    def log_base(self):
        return np.log(self.base)

    def log_partition(self):
        """composite.py:360-363, gamma.py:91-92."""
        return - 0.5 * np.sum(np.log(self.kappas), axis=1)\
            + np.sum(gammaln(self.alphas) - self.alphas * np.log(self.betas), axis=1)

    def expected_statistics(self):
        """E[lambda mu], E[-1/2 lambda mu^2], E[1/2 log lambda], E[-1/2 lambda] (composite.py:371-382)."""
        E_lmbdas = self.alphas / self.betas
        E_lmbdas_mu = E_lmbdas * self.mus
        return (E_lmbdas_mu, - 0.5 * (1. / self.kappas + self.mus * E_lmbdas_mu),
                0.5 * (digamma(self.alphas) - np.log(self.betas)), - 0.5 * E_lmbdas)

    def canonical_expected(self):
        """(c, b, W) of <E_q[eta_k], t(x)> + log_base with the (K, N, D) statistics [x, 1, 1, x^2] of
        gaussian.py:784-800 (bayesian.py:441-455): a diagonal W."""
        E1, E2, E3, E4 = self.expected_statistics()
        W = np.zeros((self.size, self.dim, self.dim))
        idx = np.arange(self.dim)
        W[:, idx, idx] = - 2. * E4
        return self.log_base() + np.sum(E2 + E3, axis=1), E1, W

    @staticmethod
    def _inner(nat, stats):
        return sum(np.einsum('kd,kd->k', n, s) for n, s in zip(nat, stats))

    def entropy(self):
        return self.log_partition() - self.log_base() - self._inner(self.nat_param, self.expected_statistics())

    def cross_entropy(self, other):
        return other.log_partition() - other.log_base() - self._inner(other.nat_param, self.expected_statistics())

    def log_likelihood(self, x):
        """sum_k log NG(mu_k, lambda_k) (composite.py:365-369, :507-509); x = (mus, lmbdas_diags)."""
        mus, lmbdas = x
        kl = self.kappas * lmbdas
        gauss = - 0.5 * np.sum(kl * (mus - self.mus)**2, axis=1) + 0.5 * np.sum(np.log(kl), axis=1)\
            - 0.5 * self.dim * np.log(2. * np.pi)
        gam = np.sum((self.alphas - 1.) * np.log(lmbdas) - self.betas * lmbdas, axis=1)\
            - np.sum(gammaln(self.alphas) - self.alphas * np.log(self.betas), axis=1)
        return np.sum(gauss + gam)


class TiedNormalGammas(StackedNormalGammas):
    """K Normal-Gammas whose Gamma factor is shared: alpha, beta = mean over k (composite.py:522-547).
    With `reference_setters` the pooled values are, like the stacked ones, never stored."""

    def nat_to_std(self, natparam):
        a, b, c, d = natparam
        mus = a / b
        alphas = np.mean(0.5 * (c + 1.), axis=0)
        betas = np.mean(0.5 * (d - b * mus**2), axis=0)
        return mus, b, np.array(self.size * [alphas]), np.array(self.size * [betas])
