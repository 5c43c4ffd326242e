"""CPU: the native batched conjugate update (mimo_amd/csrc/mimo_host.cpp, host-only entry points of
libmimo_hip.so) against the NumPy route of the same classes, which is pinned to the reference by the
golden-vector tests."""
import os
import time

import numpy as np
import pytest
from scipy.special import digamma

from mimo_amd import _lib
from mimo_amd.distributions import composite
from mimo_amd.distributions import StackedNormalWisharts, StackedMatrixNormalWisharts
from mimo_amd.utils.abstraction import Statistics as Stats
from conftest import rel_err


def _both(make, nat):
    """posterior built twice from the same natural parameters: native route, NumPy route."""
    outs = []
    for native in (True, False):
        old = composite.NATIVE_HOST
        composite.NATIVE_HOST = native
        try:
            p = make()
            p.nat_param = Stats([v.copy() for v in nat])
            outs.append(p)
        finally:
            composite.NATIVE_HOST = old
    return outs


def test_digamma():
    lib = _lib.load()
    x = np.concatenate([np.linspace(1e-3, 12., 400), np.logspace(1.1, 8, 100)])
    got = np.array([lib.mimo_host_digamma(float(v)) for v in x])
    assert np.max(np.abs(got - digamma(x)) / np.maximum(1., np.abs(digamma(x)))) < 2e-15


@pytest.mark.parametrize("K,D", [(1, 1), (3, 2), (5, 16), (64, 16), (130, 32), (7, 33)])
def test_normal_wishart_native_equals_numpy(K, D):
    rng = np.random.default_rng(K * 100 + D)
    A = rng.standard_normal((K, D, D))
    kappas = rng.uniform(0.5, 200., K)
    mus = rng.standard_normal((K, D))
    nat = [kappas[:, None] * mus, kappas, A @ A.transpose(0, 2, 1) + D * np.eye(D)
           + kappas[:, None, None] * np.einsum('kd,kl->kdl', mus, mus), rng.uniform(1., 5000., K)]
    nat_p, np_p = _both(lambda: StackedNormalWisharts(K, D), nat)
    assert 'canon' in nat_p._memo and 'hld' in nat_p._memo            # the native route really ran
    for a, b in zip(nat_p.params, np_p.params):
        assert rel_err(a, b) < 1e-12
    for a, b in zip(nat_p.canonical_expected(), np_p.canonical_expected()):
        assert rel_err(a, b) < 1e-12
    for a, b in zip(nat_p.expected_statistics(), np_p.expected_statistics()):
        assert rel_err(a, b) < 1e-12
    assert rel_err(nat_p.log_partition(), np_p.log_partition()) < 1e-12
    assert rel_err(nat_p.entropy(), np_p.entropy()) < 1e-9
    for a, b in zip(nat_p.nat_param, nat):
        assert np.array_equal(a, b)                                    # the assigned block is kept exactly
    prior = _random_nw(K, D, seed=K + D)                              # the bound's term: one native call
    v = nat_p.native_vlb(prior)
    assert v is not None and np_p.native_vlb(prior) is None
    want = np_p.entropy() - np_p.cross_entropy(prior)
    assert np.max(np.abs(v - want)) < 1e-9 * max(1., np.max(np.abs(np_p.entropy())), np.max(np.abs(np_p.cross_entropy(prior))))


@pytest.mark.parametrize("K,D", [(1, 1), (3, 2), (64, 16), (130, 32), (7, 33)])
def test_tied_normal_wishart_native_equals_numpy(K, D):
    """mimo_host_nw_vi_tied: the pooled Wishart block (composite.py:273-283) and the natural parameters the tied
    class reads back (composite.py:166-172), against the NumPy route."""
    from mimo_amd.distributions import TiedNormalWisharts
    rng = np.random.default_rng(K * 100 + D + 7)
    A = rng.standard_normal((K, D, D))
    kappas = rng.uniform(0.5, 200., K)
    mus = rng.standard_normal((K, D))
    nat = [kappas[:, None] * mus, kappas, A @ A.transpose(0, 2, 1) + D * np.eye(D)
           + kappas[:, None, None] * np.einsum('kd,kl->kdl', mus, mus), rng.uniform(1., 5000., K)]
    nat_p, np_p = _both(lambda: TiedNormalWisharts(K, D), nat)
    assert 'canon' in nat_p._memo and 'nat' in nat_p._memo and 'canon' not in np_p._memo
    assert np.array_equal(nat_p.psis[0], nat_p.psis[-1]) and np.array_equal(nat_p.nus[0], nat_p.nus[-1])
    for a, b in zip(nat_p.params, np_p.params):
        assert rel_err(a, b) < 1e-12
    for a, b in zip(nat_p.nat_param, np_p.nat_param):                  # the pooled block read back, not the assigned one
        assert rel_err(a, b) < 1e-12
    assert rel_err(nat_p.nat_param[2], nat[2]) > 1e-3 or K == 1
    for a, b in zip(nat_p.canonical_expected(), np_p.canonical_expected()):
        assert rel_err(a, b) < 1e-12
    for a, b in zip(nat_p.expected_statistics(), np_p.expected_statistics()):
        assert rel_err(a, b) < 1e-12
    assert rel_err(nat_p.log_partition(), np_p.log_partition()) < 1e-12
    assert rel_err(nat_p.entropy(), np_p.entropy()) < 1e-9
    prior = _random_nw(K, D, seed=K + D)
    v, want = nat_p.native_vlb(prior), np_p.entropy() - np_p.cross_entropy(prior)
    assert v is not None
    assert np.max(np.abs(v - want)) < 1e-9 * max(1., np.max(np.abs(np_p.entropy())), np.max(np.abs(np_p.cross_entropy(prior))))


@pytest.mark.parametrize("K,dy,dc", [(1, 1, 2), (6, 2, 4), (64, 4, 9), (33, 8, 17), (9, 3, 33)])
def test_matrix_normal_wishart_native_equals_numpy(K, dy, dc):
    rng = np.random.default_rng(K + 10 * dy + 100 * dc)
    A = rng.standard_normal((K, dc, dc)); Ks = A @ A.transpose(0, 2, 1) + dc * np.eye(dc)
    Ms = rng.standard_normal((K, dy, dc))
    A = rng.standard_normal((K, dy, dy)); Pinv = A @ A.transpose(0, 2, 1) + dy * np.eye(dy)
    nat = [Ms @ Ks, Ks, Pinv + Ms @ Ks @ Ms.transpose(0, 2, 1), rng.uniform(dy + 2., 3000., K)]
    nat_p, np_p = _both(lambda: StackedMatrixNormalWisharts(K, dc, dy), nat)
    assert 'canon_affine' in nat_p._memo
    for a, b in zip(nat_p.params, np_p.params):
        assert rel_err(a, b) < 1e-11
    for affine in (True, False):
        for a, b in zip(nat_p.canonical_expected(affine), np_p.canonical_expected(affine)):
            assert rel_err(a, b) < 1e-11
    for a, b in zip(nat_p.expected_statistics(), np_p.expected_statistics()):
        assert rel_err(a, b) < 1e-11
    assert rel_err(nat_p.entropy(), np_p.entropy()) < 1e-9


def test_not_positive_definite_takes_the_numpy_route():
    """A non-SPD block makes the native call report failure; the NumPy route then decides (as before)."""
    K, D = 5, 3
    nat = [np.zeros((K, D)), np.ones(K), np.stack(K * [np.eye(D)]), 4. * np.ones(K)]
    nat[2][2] = -np.eye(D)
    p = StackedNormalWisharts(K, D)
    p.nat_param = Stats(nat)
    assert 'canon' not in p._memo
    assert np.allclose(p.psis[2], -np.eye(D)) and np.allclose(p.psis[0], np.eye(D))


def test_host_routines_under_asan_ubsan(tmp_path):
    """mimo_host.cpp (the only native code that runs on the CPU) compiled with AddressSanitizer + UBSan and driven
    over K in {1..130}, D in {1..32}, affine / non-affine experts and a non-SPD block (GPU sanitizers are not
    available on the pool; this is the CPU build the brief asks to sanitise)."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "host_sanitize")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", "-mavx2", "-mfma", "-pthread", "-I", os.path.join(root, "include"),
           os.path.join(root, "tests", "host_sanitize.cpp"), os.path.join(root, "mimo_amd", "csrc", "mimo_host.cpp"),
           "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "sanitizer run ok" in run.stdout, run.stdout[-1000:] + run.stderr[-3000:]


def test_host_routines_under_thread_sanitizer(tmp_path):
    """The same driver under ThreadSanitizer: the persistent helper pool of the batched routines (several callers' worth of wake-ups,
    the group counter, the scratch each thread owns)."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "host_tsan")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-fno-omit-frame-pointer", "-mavx2", "-mfma", "-pthread",
           "-I", os.path.join(root, "include"), os.path.join(root, "tests", "host_sanitize.cpp"),
           os.path.join(root, "mimo_amd", "csrc", "mimo_host.cpp"), "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    if run.returncode != 0 and "unexpected memory mapping" in run.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory here")
    assert run.returncode == 0 and "sanitizer run ok" in run.stdout and "WARNING: ThreadSanitizer" not in run.stderr, \
        run.stdout[-1000:] + run.stderr[-3000:]


def _random_nw(K, D, seed):
    rs = np.random.default_rng(seed)
    A = rs.standard_normal((K, D, D))
    return StackedNormalWisharts(K, D, rs.standard_normal((K, D)), rs.uniform(0.5, 3., K),
                                 A @ A.transpose(0, 2, 1) / D + 0.2 * np.eye(D), D + rs.uniform(1., 30., K))


@pytest.mark.parametrize("K,D", [(1, 1), (4, 2), (37, 6), (256, 8), (64, 16), (9, 33)])
def test_native_gibbs_draw_equals_the_formulas(K, D):
    """mimo_host_nw_gibbs against NumPy with the SAME variates: Bartlett factor A (wishart.py:72-92), Lambda = T T'
    with T = chol(psi) A, mu = m + chol(kappa Lambda)^-T eps (gaussian.py:311-313), and the canonical form of the
    drawn Gaussians against StackedGaussiansWithPrecision.canonical (gaussian.py:352-354, 510-521)."""
    from mimo_amd.distributions import StackedGaussiansWithPrecision
    nw = _random_nw(K, D, 7 * K + D)
    mu, lam = nw.rvs(np.random.Generator(np.random.PCG64(5)))
    assert nw.drawn_canonical[0] is mu                                 # the native route really ran
    rng = np.random.Generator(np.random.PCG64(5))
    nt = D * (D - 1) // 2
    zz = rng.standard_normal((K, nt + D))
    z, eps = zz[:, :nt], zz[:, nt:]
    g = np.sqrt(rng.chisquare(nw.nus[:, None] - np.arange(D)[None, :]))
    A = np.zeros((K, D, D))
    ii = np.tril_indices(D, -1)
    A[:, ii[0], ii[1]] = z
    A[:, np.arange(D), np.arange(D)] = g
    T = np.linalg.cholesky(nw.psis) @ A
    lam2 = T @ T.transpose(0, 2, 1)
    L2 = np.linalg.cholesky(nw.kappas[:, None, None] * lam2)
    mu2 = nw.mus + np.linalg.solve(L2.transpose(0, 2, 1), eps[..., None])[..., 0]
    assert rel_err(lam, lam2) < 1e-12 and rel_err(mu, mu2) < 1e-11
    c2, b2, _ = StackedGaussiansWithPrecision(K, D, mu2, lam2).canonical()
    assert rel_err(nw.drawn_canonical[2], c2) < 1e-11 and rel_err(nw.drawn_canonical[3], b2) < 1e-11


def test_native_gibbs_draw_has_the_normal_wishart_law():
    """20000 draws from ONE Normal-Wishart: E[Lambda] = nu psi, E[mu] = m, cov(mu) = E[(kappa Lambda)^-1]
    = psi^-1 / (kappa (nu - D - 1))."""
    K, D = 20000, 3
    rs = np.random.default_rng(0)
    A = rs.standard_normal((D, D))
    psi = A @ A.T / D + 0.3 * np.eye(D)
    m, kappa, nu = rs.standard_normal(D), 2.5, 9.
    nw = StackedNormalWisharts(K, D, np.tile(m, (K, 1)), np.full(K, kappa), np.tile(psi, (K, 1, 1)), np.full(K, nu))
    mu, lam = nw.rvs(np.random.Generator(np.random.PCG64(11)))
    assert getattr(nw, 'drawn_canonical', None) is not None
    assert np.abs(lam.mean(axis=0) - nu * psi).max() < 0.05 * np.abs(nu * psi).max()
    assert np.abs(mu.mean(axis=0) - m).max() < 0.02
    cov = np.cov(mu.T)
    want = np.linalg.inv(psi) / (kappa * (nu - D - 1.))
    assert np.abs(cov - want).max() < 0.06 * np.abs(want).max()


def test_gibbs_step_hands_the_canonical_form_to_the_likelihood():
    """components.resample(rng=...) -> likelihood.canonical() returns the native (c, b) only while the likelihood holds
    exactly the drawn arrays; a later assignment of parameters recomputes."""
    from mimo_amd.distributions import StackedGaussiansWithNormalWisharts, StackedGaussiansWithPrecision
    from mimo_amd.engine import SuffStats
    from mimo_amd.mixtures.gmm import _component_stats
    K, D = 12, 5
    np.random.seed(3)
    prior = StackedNormalWisharts(K, D, np.zeros((K, D)), 1e-2 * np.ones(K), np.stack(K * [np.eye(D)]), (D + 2.) * np.ones(K))
    comp = StackedGaussiansWithNormalWisharts(K, D, prior, engine=object())
    rs = np.random.default_rng(1)
    X = rs.standard_normal((500, D)) + 3. * rs.integers(0, 3, size=(500, 1))
    lab = rs.integers(0, K, 500)
    R = np.eye(K)[lab].T
    S = SuffStats(R.sum(axis=1), R @ X, np.einsum('kn,nd,ne->kde', R, X, X))
    comp.resample(None, stats=_component_stats(S, comp), rng=np.random.Generator(np.random.PCG64(2)))
    assert comp.likelihood._canon is not None
    c, b, W = comp.likelihood.canonical()
    c2, b2, W2 = StackedGaussiansWithPrecision(K, D, comp.likelihood.mus.copy(), comp.likelihood.lmbdas.copy()).canonical()
    assert rel_err(c, c2) < 1e-11 and rel_err(b, b2) < 1e-11 and np.array_equal(W, W2)
    comp.likelihood.params = (comp.likelihood.mus + 1., comp.likelihood.lmbdas)
    c3, _, _ = comp.likelihood.canonical()
    assert rel_err(c3, StackedGaussiansWithPrecision(K, D, comp.likelihood.mus, comp.likelihood.lmbdas).canonical()[0]) < 1e-14
    assert np.abs(c3 - c).max() > 1e-3


def test_no_exception_crosses_the_c_boundary():
    """include/mimo_hip.h: every entry point runs inside a catch-all guard.  The fault hooks throw INSIDE the guarded
    boundary (an allocation failure, a std exception, a non-std exception); the call must come back with an error
    code and a message instead of unwinding through ctypes (which would abort the interpreter)."""
    lib = _lib.load()
    assert lib.mimo_debug_fault(None, 0) == _lib.OK
    assert lib.mimo_debug_fault(None, 1) == _lib.E_NOMEM
    assert b"out of host memory" in lib.mimo_last_error(None)
    assert lib.mimo_debug_fault(None, 2) == _lib.E_INTERNAL
    assert b"mimo_debug_fault" in lib.mimo_last_error(None)
    assert lib.mimo_debug_fault(None, 3) == _lib.E_INTERNAL
    assert b"unknown exception" in lib.mimo_last_error(None)
    assert lib.mimo_host_debug_fault(1) == _lib.E_NOMEM
    assert lib.mimo_host_debug_fault(2) == _lib.E_INTERNAL
    assert lib.mimo_host_debug_fault(3) == _lib.E_INTERNAL


def test_a_helper_thread_that_cannot_start_does_not_lose_work():
    """for_component_groups hands the component groups out through one counter: when a helper thread cannot be
    created the call finishes on the threads that did start, with the same numbers."""
    lib = _lib.load()
    K, D = 96, 24                       # enough work for helper threads
    rng = np.random.default_rng(5)
    A = rng.standard_normal((K, D, D))
    kappas = rng.uniform(0.5, 200., K)
    mus = rng.standard_normal((K, D))
    nat = [kappas[:, None] * mus, kappas, A @ A.transpose(0, 2, 1) + D * np.eye(D)
           + kappas[:, None, None] * np.einsum('kd,kl->kdl', mus, mus), rng.uniform(1., 5000., K)]
    old = composite.NATIVE_HOST
    composite.NATIVE_HOST = True
    try:
        ref = StackedNormalWisharts(K, D)
        ref.nat_param = Stats([v.copy() for v in nat])
        assert lib.mimo_host_debug_fault(4) == _lib.OK       # the next helper-thread start throws
        got = StackedNormalWisharts(K, D)
        got.nat_param = Stats([v.copy() for v in nat])
        assert 'canon' in got._memo
        for a, b in zip(got.canonical_expected(), ref.canonical_expected()):
            assert np.array_equal(a, b)
        for a, b in zip(got.params, ref.params):
            assert np.array_equal(a, b)
    finally:
        composite.NATIVE_HOST = old


def test_batched_rvs_consumes_numpy_random_like_the_component_loop():
    """The reference-order posterior draws (likelihood.params = posterior.rvs() of every mean-field iteration) batch
    their O(K D^3) algebra but take the variates from numpy.random in the reference's per-component order: same stream
    position afterwards and the same draws as the per-component formulas of the reference (wishart.py:72-92,
    composite.py:82-86, 607-611 — Cholesky of kron(K, Lambda) included)."""
    import numpy.random as npr
    import scipy.linalg as sla
    from mimo_amd.distributions.wishart import wishart_rvs
    rng = np.random.default_rng(2)
    K, D = 7, 5
    A = rng.standard_normal((K, D, D))
    nw = StackedNormalWisharts(K, D, rng.standard_normal((K, D)), rng.uniform(0.5, 3., K),
                               A @ A.transpose(0, 2, 1) / D + np.eye(D), rng.uniform(D + 1., 40., K))
    npr.seed(11)
    mus, lmbdas = nw.rvs()
    after = npr.random()
    npr.seed(11)
    for k in range(K):
        lm = wishart_rvs(nw.psis[k], nw.nus[k])
        ci = sla.inv(sla.cholesky(nw.kappas[k] * lm, lower=False))
        mu = nw.mus[k] + npr.normal(size=D).dot(ci.T)
        assert rel_err(lmbdas[k], lm) < 1e-12 and rel_err(mus[k], mu) < 1e-11
    assert npr.random() == after
    dy, dx = 3, 4
    A = rng.standard_normal((K, dx, dx)); B = rng.standard_normal((K, dy, dy))
    mnw = StackedMatrixNormalWisharts(K, dx, dy, rng.standard_normal((K, dy, dx)), A @ A.transpose(0, 2, 1) + np.eye(dx),
                                      B @ B.transpose(0, 2, 1) / dy + np.eye(dy), rng.uniform(dy + 1., 30., K))
    npr.seed(12)
    As, lmbdas = mnw.rvs()
    after = npr.random()
    npr.seed(12)
    for k in range(K):
        lm = wishart_rvs(mnw.psis[k], mnw.nus[k])
        ci = sla.inv(sla.cholesky(np.kron(mnw.Ks[k], lm), lower=False))
        aux = npr.normal(size=dy * dx).dot(ci.T)
        assert rel_err(lmbdas[k], lm) < 1e-12 and rel_err(As[k], mnw.Ms[k] + np.reshape(aux, (dy, dx), order='F')) < 1e-11
    assert npr.random() == after


@pytest.mark.parametrize("nbytes", [0, 5, 8, 8 * 1000 + 3, 8 * 3_000_001 + 7])
def test_host_checksum_is_the_position_weighted_sum_of_the_mixed_words(nbytes):
    """mimo_host_checksum (what bind() and the row-weight residency key on; include/mimo_hip.h): with m_i = w_i ^ (w_i >> 32) over the
    64-bit words (the tail zero-extended) and nw their number, (sum m_i, sum (nw - i) m_i) mod 2^64 — threaded above 8 MB with the same
    result; any single-byte edit changes it, and so does a swap of two unequal words."""
    import ctypes as C
    lib = _lib.load()
    rng = np.random.default_rng(nbytes)
    buf = rng.integers(0, 256, size=nbytes, dtype=np.uint8)
    out = (C.c_uint64 * 2)()
    assert lib.mimo_host_checksum(buf.ctypes.data_as(C.c_void_p), nbytes, out) == 0
    padded = np.zeros((nbytes + 7) // 8 * 8, dtype=np.uint8); padded[:nbytes] = buf
    words = [int(w) for w in padded.view(np.uint64)] if nbytes <= 10_000 else None
    if words is not None:                                   # exact integers, no NumPy wrap-around semantics
        mask = (1 << 64) - 1
        m = [w ^ (w >> 32) for w in words]
        assert int(out[0]) == sum(m) & mask and int(out[1]) == sum((len(m) - i) * v for i, v in enumerate(m)) & mask
    else:
        from mimo_amd.engine import _word_checksum_numpy
        assert (int(out[0]), int(out[1])) == _word_checksum_numpy(buf)
    if nbytes:
        first = (int(out[0]), int(out[1]))
        buf[nbytes // 2] ^= 1
        assert lib.mimo_host_checksum(buf.ctypes.data_as(C.c_void_p), nbytes, out) == 0
        assert (int(out[0]), int(out[1])) != first
        buf[nbytes // 2] ^= 1
    if nbytes >= 16 and not np.array_equal(buf[:8], buf[8:16]):
        buf[:8], buf[8:16] = buf[8:16].copy(), buf[:8].copy()
        assert lib.mimo_host_checksum(buf.ctypes.data_as(C.c_void_p), nbytes, out) == 0
        assert int(out[0]) == first[0] and int(out[1]) != first[1]                 # same words, another order
    assert lib.mimo_host_checksum(None, 8, out) == _lib.E_INVALID


def test_frozen_weights_are_fingerprinted_once_and_plain_arrays_on_every_call():
    """engine.FrozenWeights (what the hierarchical drivers wrap their weight vector in for the length of one call): the key is the
    one taken at construction; a plain array is hashed — every byte — whenever it is passed, so an in-place edit is seen."""
    from mimo_amd import engine as E
    w = np.random.default_rng(0).random(300_000)            # 2.4 MB: beyond the CRC range, inside the exact range
    fw = E.freeze_weights(w)
    assert E.freeze_weights(fw) is fw and E.freeze_weights(None) is None
    arr, key = E._weights_key(fw)
    assert arr is fw.array and key == fw.key and np.array_equal(np.asarray(fw), w) and len(fw) == w.shape[0]
    _, k1 = E._weights_key(w)
    assert k1 == key                                        # same content, same buffer: same identity
    w[12345] += 1e-12
    _, k2 = E._weights_key(w)
    assert k2 != k1 and E._weights_key(fw)[1] == key        # the plain array is re-hashed, the frozen one is not
    assert np.allclose(2.0 * np.asarray(fw), 2.0 * w)


def _gmm(K, D, tied, seed, gating='dirichlet'):
    from oracle_engine import OracleEngine
    from mimo_amd.distributions import (Dirichlet, CategoricalWithDirichlet, TruncatedStickBreaking,
                                        CategoricalWithStickBreaking, StackedGaussiansWithNormalWisharts,
                                        TiedNormalWisharts, TiedGaussiansWithNormalWisharts)
    from mimo_amd.mixtures import BayesianMixtureOfGaussians
    rs = np.random.default_rng(seed)
    eng = OracleEngine()
    eng.upload(rs.standard_normal((1500, D)) * 2. + rs.integers(0, 3, size=(1500, 1)))
    np.random.seed(seed)
    if gating == 'dirichlet':
        gate = CategoricalWithDirichlet(K, Dirichlet(K, rs.uniform(0.5, 3., K)))
    else:
        gate = CategoricalWithStickBreaking(K, TruncatedStickBreaking(K, np.ones(K), 2. * np.ones(K)))
    A = rs.standard_normal((D, D))
    args = (K, D, rs.standard_normal((K, D)), rs.uniform(0.01, 2., K), np.stack(K * [A @ A.T / D + np.eye(D)]),
            (D + 1.5) * np.ones(K))
    if tied:
        comp = TiedGaussiansWithNormalWisharts(K, D, TiedNormalWisharts(*args), engine=eng)
    else:
        comp = StackedGaussiansWithNormalWisharts(K, D, StackedNormalWisharts(*args), engine=eng)
    S = eng.label_stats(rs.integers(0, K, size=1500), K)
    return BayesianMixtureOfGaussians(gate, comp, engine=eng), eng, S


@pytest.mark.parametrize("K,D,tied", [(1, 1, 0), (4, 2, 0), (64, 16, 0), (7, 33, 0), (4, 2, 1), (64, 16, 1), (5, 9, 1)])
def test_fused_sweep_equals_the_step_by_step_route(K, D, tied):
    """mimo_host_gmm_vi_sweep / _bound against meanfield_update + canonical_expected + variational_lowerbound of the
    same objects on the NumPy route (which the golden vectors pin to the reference)."""
    from mimo_amd.distributions import native_sweep
    from mimo_amd.mixtures.gmm import _component_stats
    m1, _, S = _gmm(K, D, tied, seed=K + D)
    m2, _, _ = _gmm(K, D, tied, seed=K + D)
    fused = native_sweep.gmm_vi_sweep(m1.gating, m1.components, _component_stats(S, m1.components), S.gating_counts)
    assert fused is not None
    canon, bound = fused
    old = composite.NATIVE_HOST
    composite.NATIVE_HOST = False
    try:
        m2._update_from_stats(S, sample=False)
        want_canon, want_terms = m2.canonical_expected(), m2._vlb_prior_terms()
        ent = np.abs(m2.components.posterior.entropy()).sum() + np.abs(m2.components.posterior.cross_entropy(m2.components.prior)).sum()
    finally:
        composite.NATIVE_HOST = old
    for a, b in zip(canon, want_canon):
        assert rel_err(a, b) < 1e-12
    assert abs(bound() - want_terms) < 1e-10 * max(1., ent)
    for a, b in zip(m1.components.posterior.params, m2.components.posterior.params):
        assert rel_err(a, b) < 1e-12
    assert np.array_equal(m1.gating.posterior.alphas, m2.gating.posterior.alphas)
    # the objects are left as the step-by-step route leaves them: every derived quantity answers from them
    for a, b in zip(m1.components.posterior.nat_param, m2.components.posterior.nat_param):
        assert rel_err(a, b) < 1e-12
    for a, b in zip(m1.canonical_expected(), want_canon):
        assert rel_err(a, b) < 1e-12
    assert abs(m1._vlb_prior_terms() - want_terms) < 1e-10 * max(1., ent)


def test_fused_sweep_declines_what_it_does_not_cover():
    from mimo_amd.distributions import native_sweep
    from mimo_amd.mixtures.gmm import _component_stats
    m, _, S = _gmm(4, 2, 0, seed=3, gating='stick')
    assert native_sweep.gmm_vi_sweep(m.gating, m.components, _component_stats(S, m.components), S.gating_counts) is None
    m, _, S = _gmm(4, 2, 0, seed=3)
    bad = _component_stats(S, m.components)
    bad = Stats([bad[0], bad[1], -1e6 * np.abs(bad[2]), bad[3]])            # no longer positive definite
    assert native_sweep.gmm_vi_sweep(m.gating, m.components, bad, S.gating_counts) is None
    old = composite.NATIVE_HOST
    composite.NATIVE_HOST = False
    try:
        assert native_sweep.gmm_vi_sweep(m.gating, m.components, _component_stats(S, m.components), S.gating_counts) is None
    finally:
        composite.NATIVE_HOST = old


def test_the_driver_runs_the_fused_sweep(monkeypatch):
    """meanfield_coordinate_descent goes through the two native calls per iteration (and gets the same bound as without)."""
    from mimo_amd.distributions import native_sweep
    calls = []
    real = native_sweep.gmm_vi_sweep
    monkeypatch.setattr(native_sweep, "gmm_vi_sweep", lambda *a: calls.append(1) or real(*a))
    m1, eng, _ = _gmm(6, 3, 0, seed=11)
    np.random.seed(5)
    v1 = m1.meanfield_coordinate_descent(eng.Z, maxiter=6, progress_bar=False)
    assert len(calls) == len(v1) == 6
    monkeypatch.setattr(native_sweep, "gmm_vi_sweep", lambda *a: None)
    m2, eng2, _ = _gmm(6, 3, 0, seed=11)
    np.random.seed(5)
    v2 = m2.meanfield_coordinate_descent(eng2.Z, maxiter=6, progress_bar=False)
    assert np.allclose(v1, v2, rtol=1e-11, atol=0)


def _legacy_reference(nb, shapes, na):
    K, G = shapes.shape
    b, g, a = np.empty((K, nb)), np.empty((K, G)), np.empty((K, na))
    for k in range(K):
        b[k] = np.random.normal(size=nb)
        g[k] = np.random.standard_gamma(shapes[k])
        a[k] = np.random.normal(size=na)
    return b, g, a


@pytest.mark.parametrize("block", range(8))
def test_native_legacy_stream_is_numpy_randoms(block, monkeypatch):
    """mimo_host_legacy_draws against numpy.random itself, bit for bit: the variates (normal, standard_gamma above / below /
    at shape 1 and at 0) and the generator state afterwards (key, position, cached gaussian), from seeded states with and
    without a pending gaussian — 8 x 40 random block layouts."""
    from mimo_amd.distributions import wishart
    from mimo_amd.distributions.wishart import legacy_draws
    assert _lib.load().mimo_host_legacy_draws is not None
    monkeypatch.setattr(wishart, "_NATIVE_MIN_BLOCKS", 2)            # (a few blocks take the Python calls by default)
    for seed in range(40 * block, 40 * block + 40):
        rs = np.random.default_rng(seed)
        K, G, nb, na = int(rs.integers(2, 40)), int(rs.integers(0, 9)), int(rs.integers(0, 30)), int(rs.integers(0, 12))
        kind = seed % 4
        if kind == 0:
            shapes = rs.uniform(0.01, 1.0, (K, G))
        elif kind == 1:
            shapes = rs.uniform(0.5, 300., (K, G))
        elif kind == 2:
            shapes = np.where(rs.random((K, G)) < 0.3, 1.0, rs.uniform(0., 3., (K, G)))
        else:
            shapes = (rs.integers(1, 60, (K, 1)) - np.arange(G)[None, :] * 0.5).clip(0.0)
        outs = []
        for fn in (_legacy_reference, legacy_draws):
            np.random.seed(seed)
            if seed % 3 == 0:
                np.random.normal()                     # leaves the second variate of the pair cached
            got = fn(nb, shapes, na)
            st = np.random.get_state()
            outs.append((got, st, np.random.random(3), np.random.normal(size=2)))
        (w, sw, uw, nw), (g, sg, ug, ng) = outs
        assert all(np.array_equal(x, y) for x, y in zip(w, g)), seed
        assert np.array_equal(sw[1], sg[1]) and sw[2:] == sg[2:] and np.array_equal(uw, ug) and np.array_equal(nw, ng), seed


def test_reference_order_draws_are_the_per_component_calls():
    """The two identities the batched draws rest on, and the Bartlett variates against the reference's own call sequence
    (wishart.py:72-92: normal(n_tril), then chisquare(nu - i, size=1) per i; then the caller's normal(extra))."""
    from mimo_amd.distributions.wishart import bartlett_variates_in_reference_order
    np.random.seed(5)
    a = [np.random.chisquare(17.3 - np.arange(8)) for _ in range(4)]
    np.random.seed(5)
    b = [2.0 * np.random.standard_gamma((17.3 - np.arange(8)) / 2.0) for _ in range(4)]
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    np.random.seed(5)
    a = [np.random.gamma(np.array([0.3, 2., 7.]), np.array([.5, 3., 1.7])) for _ in range(4)]
    np.random.seed(5)
    b = [np.array([.5, 3., 1.7]) * np.random.standard_gamma(np.array([0.3, 2., 7.])) for _ in range(4)]
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    K, D, extra = 60, 5, 7                               # (60 blocks: the native route)
    nus = np.random.default_rng(1).uniform(D + 0.5, 40., K)
    np.random.seed(11)
    lower, diag, eps = bartlett_variates_in_reference_order(nus, D, extra)
    state = np.random.get_state()
    np.random.seed(11)
    for k in range(K):
        assert np.array_equal(lower[k], np.random.normal(size=D * (D - 1) // 2))
        assert np.array_equal(diag[k], np.array([(np.random.chisquare(nus[k] - i, size=1) ** 0.5)[0] for i in range(D)]))
        assert np.array_equal(eps[k], np.random.normal(size=extra))
    assert np.array_equal(state[1], np.random.get_state()[1]) and state[2:] == np.random.get_state()[2:]


@pytest.mark.parametrize("K,D,nb_iter", [(1, 1, 1), (4, 2, 5), (64, 16, 5), (7, 9, 25), (50, 3, 2)])
def test_hierarchical_update_native_equals_numpy(K, D, nb_iter):
    """mimo_host_hier_vi against the NumPy rounds of TiedGaussiansWithHierarchicalNormalWisharts.meanfield_update
    (bayesian.py:661-689 of the reference, pinned by the hierarchical golden traces)."""
    from mimo_amd.distributions import (NormalWishart, TiedGaussiansWithScaledPrecision,
                                        TiedGaussiansWithHierarchicalNormalWisharts)
    rs = np.random.default_rng(K * 31 + D)
    A = rs.standard_normal((D, D))
    outs = []
    for native in (True, False):
        np.random.seed(3)
        hyper = NormalWishart(D, rs.standard_normal(D) * 0 + 0.3, 0.7, A @ A.T / D + np.eye(D), D + 2.5)
        prior = TiedGaussiansWithScaledPrecision(K, D, kappas=np.linspace(0.01, 2., K))
        m = TiedGaussiansWithHierarchicalNormalWisharts(K, D, hyper, prior)
        X = np.random.default_rng(5).standard_normal((40 * K, D)) * 1.5 + 0.5
        lab = np.random.default_rng(6).integers(0, K, size=len(X))
        R = np.zeros((K, len(X))); R[lab, np.arange(len(X))] = 1.
        xk = R @ X
        nk = R.sum(axis=1)
        xxT = np.einsum('kn,nd,nl->kdl', R, X, X)
        old = composite.NATIVE_HOST
        composite.NATIVE_HOST = native
        try:
            m.meanfield_update(None, stats=Stats([xk, nk, xxT, nk]), nb_iter=nb_iter)
            m.meanfield_update(None, stats=Stats([xk, nk, xxT, nk]), nb_iter=nb_iter)      # (starts from the previous hyper-posterior)
        finally:
            composite.NATIVE_HOST = old
        outs.append((m.posterior.mus, m.posterior.kappas) + tuple(m.hyper_posterior.params) + tuple(m.canonical_expected())
                    + (m.likelihood.mus, m.likelihood.lmbdas))
    for a, b in zip(*outs):
        assert rel_err(np.asarray(a, dtype=float), np.asarray(b, dtype=float)) < 1e-11


def test_native_legacy_stream_in_place_from_arbitrary_generator_positions(monkeypatch):
    """The in-place route (mimo_host_legacy_draws_inplace on numpy's own key; the cached gaussian handed back by letting numpy draw
    its pair again) from generator positions all over the 624-word cycle — refills inside the draws, inside the last pair, cached
    gaussians going in and coming out — 600 layouts against numpy.random itself: variates and state."""
    from mimo_amd.distributions import wishart
    from mimo_amd.distributions.wishart import legacy_draws
    monkeypatch.setattr(wishart, "_NATIVE_MIN_BLOCKS", 1)
    assert wishart._numpy_mt() is not None
    for seed in range(600):
        rs = np.random.default_rng(10_000 + seed)
        K, G, nb, na = int(rs.integers(1, 30)), int(rs.integers(0, 6)), int(rs.integers(0, 9)), int(rs.integers(0, 5))
        kind = seed % 5
        shapes = (rs.uniform(0.01, 1.0, (K, G)) if kind == 0 else rs.uniform(0.5, 30., (K, G)) if kind == 1 else
                  np.where(rs.random((K, G)) < 0.3, 1.0, rs.uniform(0., 3., (K, G))) if kind == 2 else
                  rs.uniform(1.0, 1.2, (K, G)) if kind == 3 else (rs.integers(1, 60, (K, 1)) - np.arange(G)[None, :] * 0.5).clip(0.0))
        pre = int(rs.integers(0, 700))
        outs = []
        for fn in (_legacy_reference, legacy_draws):
            np.random.seed(seed)
            np.random.random(pre)
            if seed % 3 == 0:
                np.random.normal()
            got = fn(nb, shapes, na)
            outs.append((got, np.random.get_state(), np.random.random(2), np.random.normal(size=3)))
        (w, sw, uw, nw), (g, sg, ug, ng) = outs
        assert all(np.array_equal(x, y) for x, y in zip(w, g)), seed
        assert np.array_equal(sw[1], sg[1]) and sw[2:] == sg[2:] and np.array_equal(uw, ug) and np.array_equal(nw, ng), seed


def test_helper_pool_survives_fork_and_concurrent_callers():
    """The persistent helper threads of the batched routines (mimo_host.cpp: HelperPool): a child process created by fork() after the
    pool exists has none of its threads and starts its own; two Python threads inside the routines at once get the same numbers (the
    second finds the pool busy and runs alone)."""
    import threading
    K, D = 128, 32                                   # enough work for the pool
    rng = np.random.default_rng(8)
    A = rng.standard_normal((K, D, D))
    kappas = rng.uniform(0.5, 200., K)
    mus = rng.standard_normal((K, D))
    nat = [kappas[:, None] * mus, kappas, A @ A.transpose(0, 2, 1) + D * np.eye(D)
           + kappas[:, None, None] * np.einsum('kd,kl->kdl', mus, mus), rng.uniform(1., 5000., K)]

    def solve():
        p = StackedNormalWisharts(K, D)
        assert p._assign_native([v.copy() for v in nat])
        return p.canonical_expected()[2]

    old = composite.NATIVE_HOST
    composite.NATIVE_HOST = True
    try:
        ref = solve()
        out = [None] * 4
        ts = [threading.Thread(target=lambda i=i: out.__setitem__(i, solve())) for i in range(4)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert all(np.array_equal(o, ref) for o in out)
        pid = os.fork()
        if pid == 0:                                # child: the parent's helpers are gone
            code = 1
            try:
                code = 0 if np.array_equal(solve(), ref) and np.array_equal(solve(), ref) else 2
            finally:
                os._exit(code)
        deadline = time.time() + 60
        while time.time() < deadline:
            done, status = os.waitpid(pid, os.WNOHANG)
            if done:
                break
            time.sleep(0.05)
        else:
            os.kill(pid, 9)
            raise AssertionError("the forked child hung in the batched routine")
        assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, status
    finally:
        composite.NATIVE_HOST = old
