"""CPU: host logic of mimo_amd (canonical forms, conjugate updates, drivers) against the reference's
golden vectors, with the oracle-backed engine double standing in for the GPU (tests/oracle_engine.py)."""
import numpy as np
import pytest

from conftest import GMM_CASES, ILR_CASES, GIBBS_CASES
from oracle_engine import OracleEngine
import model_checks as mc


@pytest.mark.parametrize("name", GMM_CASES)
def test_gmm_tables_stats_elbo(name):
    mc.check_gmm_case(name, OracleEngine())


@pytest.mark.parametrize("name", GMM_CASES)
def test_gmm_vi_trace(name):
    mc.check_gmm_vi_trace(name, OracleEngine())


@pytest.mark.parametrize("name", GIBBS_CASES)
def test_gibbs_trace(name):
    mc.check_gibbs_trace(name, OracleEngine())


@pytest.mark.parametrize("name", ILR_CASES)
def test_ilr_tables_stats_elbo(name):
    mc.check_ilr_case(name, OracleEngine())


@pytest.mark.parametrize("name", ILR_CASES)
def test_ilr_vi_trace(name):
    mc.check_ilr_vi_trace(name, OracleEngine())


def test_em_map_svi_driver_traces():
    mc.check_driver_traces("drivers_d3_k5_dir", OracleEngine())


def test_ilr_scaled_gibbs_then_svi():
    mc.check_ilr_svi("ilr_svi_dx2_dy1_k8", OracleEngine())


def test_tied_gmm_gibbs_vi_em():
    mc.check_tied_gmm("tied_gmm_d3_k5", OracleEngine())


@pytest.mark.parametrize("name", ["diag_gmm_d3_k5", "tied_diag_gmm_d4_k6"])
def test_diag_gmm_all_drivers(name):
    mc.check_diag_gmm(name, OracleEngine())


@pytest.mark.parametrize("name", ["hier_gmm_d2_k4_m2", "hier_gmm_d3_k3_m3"])
def test_hierarchical_gmm_all_drivers(name):
    mc.check_hier_gmm(name, OracleEngine())


@pytest.mark.parametrize("name", ["hier_ilr_dx1_dy1_k3", "hier_ilr_dx2_dy2_k4"])
def test_hierarchical_ilr_tied_activation(name):
    mc.check_hier_ilr(name, OracleEngine())


@pytest.mark.parametrize("name", ["tied_ilr_sine_k8", "tied_ilr_dx3_dy2_k6"])
def test_tied_ilr_flow_and_prediction(name):
    mc.check_tied_ilr_prediction(name, OracleEngine())


def test_batched_samplers_match_the_reference_law():
    """The batched (Generator) Normal-Wishart / Matrix-Normal-Wishart samplers of the fast Gibbs path
    have the same first two moments as the reference-order samplers."""
    import numpy as np
    import numpy.random as npr
    from mimo_amd.distributions import StackedNormalWisharts, StackedMatrixNormalWisharts
    rng = np.random.default_rng(3)
    K, D = 4000, 3
    A = rng.standard_normal((D, D)); psi = A @ A.T / D + np.eye(D)
    nw = StackedNormalWisharts(K, D, np.tile([1., -2., .5], (K, 1)), 2.5 * np.ones(K), np.tile(psi, (K, 1, 1)), 7.5 * np.ones(K))
    mus_f, lm_f = nw.rvs(rng)
    npr.seed(5); mus_r, lm_r = nw.rvs()
    assert np.allclose(lm_f.mean(0), 7.5 * psi, rtol=0.05, atol=0.4) and np.allclose(lm_r.mean(0), lm_f.mean(0), rtol=0.05, atol=0.6)
    assert np.allclose(mus_f.mean(0), [1., -2., .5], atol=0.03)
    assert np.allclose(np.cov(mus_f.T), np.cov(mus_r.T), rtol=0.25, atol=0.01)
    dy, dc = 2, 3
    Kc = np.array([[2., .3, 0.], [.3, 1., .2], [0., .2, 3.]])
    mnw = StackedMatrixNormalWisharts(K, dc, dy, np.zeros((K, dy, dc)), np.tile(Kc, (K, 1, 1)),
                                      np.tile(np.eye(dy), (K, 1, 1)), 6. * np.ones(K))
    A_f, _ = mnw.rvs(np.random.default_rng(4))
    npr.seed(6); A_r, _ = mnw.rvs()
    cf = np.cov(A_f.reshape(K, -1).T); cr = np.cov(A_r.reshape(K, -1).T)
    # both against the exact law: cov(vec A) = E[Lambda^-1] (x) K^-1, E[Lambda^-1] = psi^-1 / (nu - dy - 1)
    exact = np.kron(np.eye(dy) / (6. - dy - 1.), np.linalg.inv(Kc))
    assert np.allclose(cf, exact, rtol=0.2, atol=0.02) and np.allclose(cr, exact, rtol=0.2, atol=0.02)


def test_bind_sees_in_place_edits():
    """engine.bind() keys the resident copy on address + shape + a content fingerprint: editing the bound array in
    place (the reference re-reads its arguments on every call) uploads it again; an untouched array is not re-sent."""
    import numpy as np
    from mimo_amd import engine as E
    from mimo_amd.distributions.lingauss import joint_rows

    class Counting(OracleEngine):
        uploads = 0

        def upload(self, Z):
            Counting.uploads += 1
            super().upload(np.array(Z, copy=True))       # a device copy would not follow host edits either

    rng = np.random.default_rng(0)
    for shape in ((500, 3), (400_000, 2)):               # fully hashed / strided-sample fingerprint
        eng, X = Counting(), rng.standard_normal(shape)
        Counting.uploads = 0
        E.bind(eng, X); E.bind(eng, X)
        assert Counting.uploads == 1
        X -= X.mean(axis=0)                               # centring in place
        E.bind(eng, X)
        assert Counting.uploads == 2 and np.array_equal(eng.Z, X)
        X[len(X) // 3: 2 * len(X) // 3] = 0.              # a contiguous block overwritten
        E.bind(eng, X)
        assert Counting.uploads == 3 and np.array_equal(eng.Z, X)
    x, y = rng.standard_normal((300, 2)), rng.standard_normal((300, 1))
    z0 = joint_rows(x, y)
    assert joint_rows(x, y) is z0
    y *= 2.
    z1 = joint_rows(x, y)
    assert z1 is not z0 and np.array_equal(z1[:, 2:], y)


def test_bind_sees_row_swaps_and_single_elements():
    """The content key is position dependent: an in-place swap of two rows of a 2 - 16 MB array — same bytes,
    another order — uploads again, and so does one edited element; the native checksum and its NumPy form are one function."""
    import numpy as np
    from mimo_amd import engine as E

    class Counting(OracleEngine):
        uploads = 0

        def upload(self, Z):
            Counting.uploads += 1
            super().upload(np.array(Z, copy=True))

    rng = np.random.default_rng(1)
    for n in (0, 5, 8, 13, 4096, 100_003):
        w = rng.integers(0, 256, size=n, dtype=np.uint8)
        assert E._word_checksum(w) == E._word_checksum_numpy(w) and len(E._word_checksum(w)) == 2
    X = np.ones((3 * 4096, 16)); X[7] = 2.
    a = E._word_checksum(X)
    X[[7, 7 + 2048]] = X[[7 + 2048, 7]]                  # "round" doubles 2^11 rows apart: the weighted sum of the raw words would not move
    assert E._word_checksum(X) != a
    eng, X = Counting(), rng.standard_normal((400_000, 4))           # 12.8 MB: every word is summed
    E.bind(eng, X); E.bind(eng, X)
    assert Counting.uploads == 1
    X[[11, 300_000]] = X[[300_000, 11]]
    E.bind(eng, X)
    assert Counting.uploads == 2 and np.array_equal(eng.Z, X)
    rng.shuffle(X)
    E.bind(eng, X)
    assert Counting.uploads == 3 and np.array_equal(eng.Z, X)
    X[123_456, 2] += 1e-9
    E.bind(eng, X)
    assert Counting.uploads == 4 and np.array_equal(eng.Z, X)


def test_bound_rows_are_verified_behind_the_first_pass():
    """Arrays above 16 MB: bind() compares a sample and, when it matches, checks EVERY byte on a helper thread while the first
    pass runs (engine.BoundDataGuard); a single edited element that the sample misses is found before results leave the
    engine: the array is uploaded again, a warning raised and the pass repeated — synchronous and asynchronous form."""
    import numpy as np
    from mimo_amd import engine as E

    class Guarded(E.BoundDataGuard, OracleEngine):
        uploads = 0

        def upload(self, Z):
            Guarded.uploads += 1
            self._verify = None
            self._uploaded_sum = E._word_checksum(np.ascontiguousarray(Z, dtype=float))
            super().upload(np.array(Z, copy=True))

        def data_checksum(self):
            return self._uploaded_sum

        estep = E.checked(OracleEngine.estep)

    rng = np.random.default_rng(2)
    N, D, K = 1_100_000, 2, 3                                         # 17.6 MB
    X = rng.standard_normal((N, D))
    c, b, W = rng.standard_normal(K), rng.standard_normal((K, D)), np.stack(K * [np.eye(D)])
    eng = Guarded()
    E.bind(eng, X)
    S0, _ = eng.estep(c, b, W)
    assert Guarded.uploads == 1 and eng._verify is None
    E.bind(eng, X)                                                    # untouched: verified, no upload, no warning
    assert eng._verify is not None
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        S1, _ = eng.estep(c, b, W)
    assert Guarded.uploads == 1 and eng._verify is None and np.array_equal(S1.sxx, S0.sxx)
    X[777_777, 1] += 5.                                               # one element: the strided sample does not see it
    assert E._bind_key(X) == eng._bound_key
    E.bind(eng, X)
    with pytest.warns(RuntimeWarning, match="edited in place"):
        S2, _ = eng.estep(c, b, W)
    assert Guarded.uploads == 2 and np.array_equal(eng.Z, X) and not np.array_equal(S2.sxx, S0.sxx)
    from oracle import mimo_oracle as O
    from scipy.special import logsumexp
    L = O.canonical_eval(X, c, b, W)
    n, sx, sxx = O.packed_stats(X, np.exp(L - logsumexp(L, axis=0)))
    assert np.allclose(S2.sxx, sxx, rtol=1e-12)
    E.bind(eng, X)                                                    # and now it is clean again
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        eng.estep(c, b, W)
    assert Guarded.uploads == 2


def test_sample_discrete_from_log_and_random_start_on_the_double():
    """Host logic of the two API additions with the oracle-backed double: axis handling / numpy.random call shape of
    sample_discrete_from_log, and the init_rng switch of the drivers."""
    import numpy as np
    import numpy.random as npr
    from oracle import mimo_oracle as O
    from mimo_amd.utils.stats import sample_discrete_from_log
    rng = np.random.default_rng(0)
    p = rng.standard_normal((6, 50, 3))
    npr.seed(1)
    got = sample_discrete_from_log(p, axis=0, engine=OracleEngine())
    npr.seed(1)
    u = npr.random(size=(1, 50, 3))
    assert np.array_equal(got.reshape(-1), O.sample_discrete_from_log(p.reshape(6, -1), u.reshape(-1))) and got.shape == (50, 3)
    from conftest import load_golden
    g = load_golden("gmm_c1_d2_k4_dir")
    kind, model = mc.build_gmm(g, OracleEngine())
    vlb = model.meanfield_coordinate_descent(g["X"], randomize=True, maxiter=5, tol=0., progress_bar=False, init_rng='philox', seed=3)
    assert np.all(np.diff(vlb) > -1e-8 * abs(vlb[-1]))
    with pytest.raises(ValueError):
        model.meanfield_coordinate_descent(g["X"], randomize=True, maxiter=1, progress_bar=False, init_rng='nope')


@pytest.mark.parametrize("name", ["nan_rows_gmm_d3_k5", "nan_rows_gmm_d16_k70"])
def test_rows_with_nan(name):
    mc.check_nan_rows(name, OracleEngine())


def test_rows_with_nan_in_a_linear_gaussian_mixture():
    """the reference's element-wise NaN rules of the experts' density (lingauss.py:150-151, ilr.py:71-75), host classes over the
    oracle-backed engine double against outputs of the reference"""
    mc.check_nan_rows_ilr("nan_rows_ilr_dx2_dy1_k6", OracleEngine())


@pytest.mark.parametrize("n,k", [(4_000_000, 4096), (10_000_000, 128), (5000, 4096), (4097, 4096), (1 << 31, 1000),
                                 (100000, 99990), (1000, 70), (300, 64), (20000, 4096), ((1 << 32) - 1, 5000),
                                 (1 << 32, 5000), (50, 10), (4_000_000, 5), (300, 300), (257, 256), (70000, 17000), (16405, 4096),
                                 (16406, 4096), (1000, 999)])
def test_sample_indices_is_random_sample(n, k):
    """utils.data.sample_indices (the minibatch draw of the SVI drivers, data.py:9-12 of the reference) returns what
    random.sample(range(n), k) returns and leaves Python's generator in the same state — on the NumPy block route and on
    the fallbacks alike."""
    import random
    from mimo_amd.utils.data import sample_indices
    for seed in (0, 1, 12345):
        random.seed(seed)
        random.gauss(0., 1.)                            # (a pending gauss_next value must survive the state round trip)
        want = [random.sample(range(n), k) for _ in range(3)]
        state, nxt = random.getstate(), (random.gauss(0., 1.), random.random())
        random.seed(seed)
        random.gauss(0., 1.)
        got = [sample_indices(n, k) for _ in range(3)]
        assert got == want and all(type(i) is int for i in got[0])
        assert random.getstate() == state and (random.gauss(0., 1.), random.random()) == nxt


class DeferredOracleEngine(OracleEngine):
    """OracleEngine with HipEngine's asynchronous pair, evaluated at estep_wait(): the drivers take their pipelined branches
    (the fused pass launched before the host algebra that overlaps it; the SVI loop's reordered queue), and a result that is
    read before it was waited for, or a pass launched with the wrong iteration's parameters, shows up against the goldens."""

    def spawn(self):
        return DeferredOracleEngine()

    def estep_async(self, c, b, W, row_weights=None, stats=True):
        assert getattr(self, "_pending", None) is None, "a second pass was launched on a context that has one in flight"
        self._pending = (np.array(c), np.array(b), np.array(W), row_weights, stats)

    def estep_wait(self):
        c, b, W, w, stats = self._pending
        self._pending = None
        kw = {} if w is None else {"row_weights": w}
        return self.estep(c, b, W, stats=stats, **kw)


def test_pipelined_drivers_reproduce_the_reference_traces():
    """The golden VI / SVI traces (generated by the reference) through the asynchronous branches of the drivers."""
    mc.check_gmm_case("gmm_c1_d2_k4_dir", DeferredOracleEngine())
    mc.check_driver_traces("drivers_d3_k5_dir", DeferredOracleEngine())
    mc.check_ilr_svi("ilr_svi_dx2_dy1_k8", DeferredOracleEngine())
    mc.check_tied_gmm("tied_gmm_d3_k5", DeferredOracleEngine())
