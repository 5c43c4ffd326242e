"""Wishart distribution (host-side, O(D^3)) and batched helpers shared by the Normal-Wishart and
Matrix-Normal-Wishart blocks.  Reference: mimo/distributions/wishart.py:11-153."""
import ctypes as C

import numpy as np
import numpy.random as npr
from scipy.special import multigammaln, digamma

from mimo_amd.utils.abstraction import Statistics as Stats


def sum_log_diag_chol(mats):
    """sum_i log diag(chol(A))_i for a stack (..., D, D) of SPD matrices (= 1/2 logdet A)."""
    chol = np.linalg.cholesky(mats)
    return np.sum(np.log(np.diagonal(chol, axis1=-2, axis2=-1)), axis=-1)


def wishart_log_partition(psis, nus, half_logdet=None):
    """wishart.py:129-132, vectorised over a stack: nu D/2 log 2 + ln Gamma_D(nu/2) + nu sum log diag chol psi.
    `half_logdet` = sum log diag chol psi if the caller already has it."""
    D = psis.shape[-1]
    hld = sum_log_diag_chol(psis) if half_logdet is None else half_logdet
    return 0.5 * nus * D * np.log(2) + multigammaln(nus / 2., D) + nus * hld


def wishart_expected_logdet(psis, nus, half_logdet=None):
    """wishart.py:139-143: E[logdet X] = sum_i psi((nu - i)/2) + D log 2 + 2 sum log diag chol psi."""
    D = psis.shape[-1]
    i = np.arange(D)
    hld = sum_log_diag_chol(psis) if half_logdet is None else half_logdet
    return np.sum(digamma((np.asarray(nus)[..., None] - i) / 2.), axis=-1) + D * np.log(2.) + 2. * hld


def wishart_rvs(psi, nu):
    """Bartlett decomposition with the reference's RNG call order (wishart.py:72-92):
    normal(n_tril) for the strict lower triangle, then D separate chisquare(nu - i) draws."""
    D = psi.shape[0]
    n_tril = D * (D - 1) // 2
    covariances = npr.normal(size=n_tril).reshape((n_tril,))
    # (array ** 0.5 as in the reference: numpy turns it into sqrt — correctly rounded; a float64 SCALAR ** 0.5 goes to libm's pow
    # and differs from it in the last bit once in a few hundred values)
    variances = np.array([(npr.chisquare(nu - i, size=1) ** 0.5)[0] for i in range(D)])
    A = np.zeros((D, D))
    A[np.tril_indices(D, k=-1)] = covariances
    A[np.diag_indices(D)] = variances
    T = np.linalg.cholesky(psi) @ A
    return T @ T.T


_NATIVE_MIN_BLOCKS = 4


def legacy_draws(n_before, shapes, n_after):
    """Per block k of `shapes` (K, G): numpy.random.normal(size=n_before), numpy.random.standard_gamma(shapes[k]) (G values, in
    order), numpy.random.normal(size=n_after) — the variates and the final state of numpy.random's global generator are those of
    the 3 K Python calls; computed by mimo_host_legacy_draws_inplace (a restatement of numpy's legacy MT19937 stream, working on the
    bit generator's own key: 5 us of hand-over) when the library is there and the global generator is the stock one — else by
    mimo_host_legacy_draws through get_state / set_state (0.1 ms: numpy converts the key element by element), else by the calls
    themselves.  Not atomic against other threads drawing from numpy.random meanwhile — like the loop it replaces.
    -> (before (K, n_before), gammas (K, G), after (K, n_after))."""
    shapes = np.ascontiguousarray(shapes, dtype=np.float64)
    K, G = shapes.shape
    before, gam, after = np.empty((K, n_before)), np.empty((K, G)), np.empty((K, n_after))
    lib = _native_lib() if K >= _NATIVE_MIN_BLOCKS else None
    if lib is not None and np.all(shapes >= 0.):
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        mt = _numpy_mt()
        if mt is not None and (n_before or n_after or np.any(shapes > 1.)):       # (at least one gaussian will be drawn)
            # numpy's own state, in place: empty RandomState's gaussian cache (learning what was in it), draw natively on the bit
            # generator's key, hand a gaussian that is left over back by letting numpy draw its pair again (mimo_hip.h)
            rs, addr = mt
            state = _MTState.from_address(addr)
            saved, pos0, word0 = C.string_at(addr, C.sizeof(_MTState)), state.pos, state.key[0]
            g0 = rs.standard_normal()
            if state.pos == pos0 and state.key[0] == word0:
                had = 1                                   # nothing was consumed: g0 was the cached value, the cache is empty now
            else:
                rs.standard_normal()                      # a pair was drawn: take its second half out of the cache, ...
                C.memmove(addr, saved, len(saved))        # ... and put the generator back where it stood
                had, g0 = 0, 0.0
            redraw, final = C.c_int(0), _MTState()
            rc = lib.mimo_host_legacy_draws_inplace(addr, addr + _MTState.pos.offset, had, g0, K, n_before, G, n_after,
                                                    ptr(shapes), ptr(before), ptr(gam), ptr(after), C.byref(redraw), C.addressof(final))
            if rc == 0:
                if redraw.value:
                    rs.standard_normal()                  # numpy draws the last pair again and keeps its second half ...
                    C.memmove(addr, C.addressof(final), C.sizeof(_MTState))     # ... then stands where the draws ended
                return before, gam, after
            C.memmove(addr, saved, len(saved))            # (never seen: back to the state found, cached gaussian included, ...
            if had:                                       # ... and to the copying route below)
                st = npr.get_state()
                npr.set_state((st[0], st[1], st[2], 1, g0))
        st = npr.get_state()
        if st[0] == 'MT19937':
            key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
            pos, has, g = C.c_int(int(st[2])), C.c_int(int(st[3])), C.c_double(float(st[4]))
            if lib.mimo_host_legacy_draws(ptr(key), C.byref(pos), C.byref(has), C.byref(g), K, n_before, G, n_after,
                                          ptr(shapes), ptr(before), ptr(gam), ptr(after)) == 0:
                npr.set_state(('MT19937', key, pos.value, has.value, g.value))
                return before, gam, after
    for k in range(K):
        before[k] = npr.normal(size=n_before)
        gam[k] = npr.standard_gamma(shapes[k])
        after[k] = npr.normal(size=n_after)
    return before, gam, after


class _MTState(C.Structure):
    """numpy/random/src/mt19937/mt19937.h: mt19937_state."""
    _fields_ = [("key", C.c_uint32 * 624), ("pos", C.c_int)]


def _numpy_mt():
    """(numpy.random's global RandomState, address of its bit generator's MT19937 state) while both are the stock ones, else None."""
    rs = getattr(npr.mtrand, '_rand', None)
    bg = getattr(rs, '_bit_generator', None)
    if type(rs) is not npr.RandomState or type(bg) is not np.random.MT19937:
        return None
    try:
        return rs, int(bg.ctypes.state_address)
    except Exception:
        return None


def _native_lib():
    from mimo_amd.distributions import composite
    return composite._native()


def bartlett_variates_in_reference_order(nus, D, extra):
    """The variates of K Bartlett draws, each followed by `extra` standard normals, taken from numpy.random in the
    order the reference consumes them per component (wishart.py:72-92, then the caller's normal(size=extra)):
    normal(D(D-1)/2), chisquare(nu - i) for i < D, normal(extra).  chisquare(df) IS 2 standard_gamma(df / 2) in the legacy
    generator (both operations exact in float64), and a vector of shapes is walked in order, so the K blocks are one
    `legacy_draws` call (tests/test_host_native.py holds the stream against the per-call form).
    Returns (lower (K, n_tril), diag (K, D), eps (K, extra))."""
    K, n_tril = len(nus), D * (D - 1) // 2
    if K < _NATIVE_MIN_BLOCKS:            # a handful of blocks: the calls themselves (no wrapper in front of a 0.1 ms sweep)
        lower, diag, eps = np.empty((K, n_tril)), np.empty((K, D)), np.empty((K, extra))
        dof_off = np.arange(D)
        for k in range(K):
            lower[k] = npr.normal(size=n_tril)
            diag[k] = npr.chisquare(nus[k] - dof_off)
            eps[k] = npr.normal(size=extra)
        return lower, np.sqrt(diag), eps
    dof = np.asarray(nus, dtype=float)[:, None] - np.arange(D)[None, :]
    lower, gam, eps = legacy_draws(n_tril, dof / 2.0, extra)
    return lower, np.sqrt(2.0 * gam), eps


def wishart_from_bartlett(psis, lower, diag):
    """Lambda_k = T T', T = chol(psi_k) A_k, for a stack of Bartlett factors given by their variates."""
    K, D = diag.shape
    A = np.zeros((K, D, D))
    ti = np.tril_indices(D, k=-1)
    A[:, ti[0], ti[1]] = lower
    A[:, np.arange(D), np.arange(D)] = diag
    T = np.linalg.cholesky(psis) @ A
    return T @ np.swapaxes(T, 1, 2)


def wishart_rvs_batched(psis, nus, rng):
    """K Wishart draws in one shot (Bartlett), from a numpy Generator.  Same distribution as
    wishart_rvs but NOT the reference's numpy.random call order — used by the fast Gibbs path."""
    K, D = psis.shape[0], psis.shape[-1]
    A = np.tril(rng.standard_normal((K, D, D)), k=-1)
    dof = np.asarray(nus)[:, None] - np.arange(D)[None, :]
    A[:, np.arange(D), np.arange(D)] = np.sqrt(rng.chisquare(dof))
    T = np.linalg.cholesky(psis) @ A
    return T @ np.swapaxes(T, 1, 2)


class Wishart:

    def __init__(self, dim, psi=None, nu=None):
        self.dim = dim
        self.psi = psi
        self.nu = nu

    @property
    def params(self):
        return self.psi, self.nu

    @params.setter
    def params(self, values):
        self.psi, self.nu = values

    @property
    def nat_param(self):
        return self.std_to_nat(self.params)

    @nat_param.setter
    def nat_param(self, natparam):
        self.params = self.nat_to_std(natparam)

    @staticmethod
    def std_to_nat(params):
        a = - 0.5 * np.linalg.inv(params[0])
        return Stats([a, 0.5 * (params[1] - a.shape[0] - 1)])

    @staticmethod
    def nat_to_std(natparam):
        psi = - 0.5 * np.linalg.inv(natparam[0])
        return psi, 2. * natparam[1] + psi.shape[0] + 1

    @property
    def psi_chol(self):
        return np.linalg.cholesky(self.psi)

    def mean(self):
        return self.nu * self.psi

    def mode(self):
        assert self.nu >= (self.dim + 1)
        return (self.nu - self.dim - 1) * self.psi

    def rvs(self, size=1):
        return wishart_rvs(self.psi, self.nu)

    @property
    def base(self):
        return 1.

    def log_base(self):
        return np.log(self.base)

    def log_partition(self):
        return wishart_log_partition(self.psi, self.nu)

    def log_likelihood(self, x):
        log_lik = 0.5 * (self.nu - self.dim - 1) * np.linalg.slogdet(x)[1]\
            - 0.5 * np.trace(np.linalg.solve(self.psi, x))
        return - self.log_partition() + self.log_base() + log_lik

    def expected_statistics(self):
        return self.nu * self.psi, wishart_expected_logdet(self.psi, self.nu)

    def entropy(self):
        nat, stats = self.nat_param, self.expected_statistics()
        return self.log_partition() - self.log_base() - (np.tensordot(nat[0], stats[0]) + nat[1] * stats[1])

    def cross_entropy(self, dist):
        nat, stats = dist.nat_param, self.expected_statistics()
        return dist.log_partition() - dist.log_base() - (np.tensordot(nat[0], stats[0]) + nat[1] * stats[1])
