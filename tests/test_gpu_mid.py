"""GPU (-m gpu): the mid kernels (mimo_mid.hip: softmax + statistics pass of K <= 32 components over Dz = 9 .. 32, row-owner
E-step waves + column-owner statistics waves) against the oracle: every Dz, both row-block counts, four- and eight-wave
workgroups, ragged / tiny / many-super-step row counts, row weights, rows with NaN, the asynchronous form, repeated launches."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _random_problem(rng, N, D, K):
    Z = rng.standard_normal((N, D)) * 2.0 + rng.standard_normal(D)
    A = rng.standard_normal((K, D, D))
    W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D)
    mu = rng.standard_normal((K, D)) * 2
    b = np.einsum('kde,ke->kd', W, mu)
    c = -0.5 * np.einsum('kd,kd->k', mu, b) + rng.standard_normal(K) * 0.1
    return Z, c, b, W


def _check(engine, Z, c, b, W, weights=None):
    from oracle import mimo_oracle as O
    from scipy.special import logsumexp
    N = len(Z)
    L = O.canonical_eval(Z, c, b, W)
    lse = logsumexp(L, axis=0) if N else np.zeros(0)
    R = np.exp(L - lse)
    n, sx, sxx = O.packed_stats(Z, R if weights is None else R * weights[None, :])
    S, sc = engine.estep(c, b, W, row_weights=weights)
    assert rel_err(S.n, n) < 1e-11 and rel_err(S.sx, sx) < 1e-11 and rel_err(S.sxx, sxx) < 1e-11
    if N:
        assert abs(sc[0] - lse.sum()) < 1e-12 * max(1., abs(lse.sum()))
    return S, sc


MID_SHAPES = [(D, K) for D in range(17, 33) for K in ((9, 32) if D % 2 else (16, 17))] + \
             [(27, 8), (32, 7), (30, 5), (24, 24), (32, 31), (20, 12), (28, 20),
              (13, 33), (16, 48), (19, 48), (20, 40), (21, 35), (24, 48), (25, 41), (26, 48), (14, 16), (12, 13),
              (8, 96), (6, 80), (9, 72), (12, 96), (14, 90), (16, 80), (10, 65), (18, 64), (21, 50), (20, 60), (9, 33), (11, 48),
              (20, 96), (23, 90), (16, 96), (24, 80), (26, 70), (28, 64), (22, 49), (32, 48), (29, 40), (27, 33),
              (12, 112), (14, 128), (20, 100), (8, 112), (10, 128)]     # three to eight row blocks; the last two lines: one wave per SIMD


@pytest.mark.parametrize("D,K", MID_SHAPES)
def test_mid_kernels_vs_oracle(engine, D, K):
    rng = np.random.default_rng(7700 + 40 * D + K)
    engine.tune("mid_narrow_k", 1)          # (few components: the narrow kernels by default where they exist)
    for N in (1, 17, 5003):
        Z, c, b, W = _random_problem(rng, N, D, K)
        engine.upload(Z)
        assert engine.plan(K)["kind"] == "mid"
        _check(engine, Z, c, b, W)
    # many super-steps per workgroup (three CUs' worth of workgroups), weights, second launch, asynchronous form
    N = 20011
    Z, c, b, W = _random_problem(rng, N, D, K)
    engine.upload(Z)
    engine.tune("num_cu", 3)
    try:
        S, sc = _check(engine, Z, c, b, W)
        S2, sc2 = engine.estep(c, b, W)
        assert np.array_equal(S2.sxx, S.sxx) and np.array_equal(S2.sx, S.sx) and sc2[0] == sc[0]
        engine.estep_async(c, b, W)
        S3, sc3 = engine.estep_wait()
        assert np.array_equal(S3.sxx, S.sxx) and sc3[0] == sc[0]
        _check(engine, Z, c, b, W, weights=rng.random(N))
    finally:
        engine.tune("num_cu", 0)
    # a component switched off by its weight (log 0) enters like a padding component
    c2 = c.copy(); c2[K // 2] = -np.inf
    if K > 1:
        from oracle import mimo_oracle as O
        from scipy.special import logsumexp
        L2 = O.canonical_eval(Z, np.where(np.isinf(c2), -1e300, c2), b, W)
        lse2 = logsumexp(L2, axis=0)
        n2, _, sxx2 = O.packed_stats(Z, np.exp(L2 - lse2))
        So, sco = engine.estep(c2, b, W)
        assert So.n[K // 2] < 1e-290 and rel_err(So.sxx, sxx2) < 1e-11 and abs(sco[0] - lse2.sum()) < 1e-12 * abs(lse2.sum())


@pytest.mark.parametrize("D,K", [(9, 9), (12, 16), (13, 32), (16, 17), (16, 32), (10, 24), (15, 12), (11, 30), (14, 16), (5, 96), (7, 40), (16, 64), (5, 17)])
def test_mid_kernels_below_their_default_range(engine, D, K):
    """Dz = 9 .. 16: the kernels exist there too (mimo_tune "mid_min_d" routes them); the router's default prefers the narrow /
    tile / row-owner kernels where they measured faster."""
    rng = np.random.default_rng(7900 + 40 * D + K)
    N = 20011
    Z, c, b, W = _random_problem(rng, N, D, K)
    engine.upload(Z)
    engine.tune("mid_min_d", 5)
    engine.tune("mid_narrow_k", 1)
    try:
        assert engine.plan(K)["kind"] == "mid"
        S, sc = _check(engine, Z, c, b, W)
        S2, sc2 = engine.estep(c, b, W)
        assert np.array_equal(S2.sxx, S.sxx) and sc2[0] == sc[0]
    finally:
        engine.tune("mid_min_d", 0)
        engine.tune("mid_narrow_k", 0)


@pytest.mark.parametrize("D,K", [(20, 16), (32, 32), (27, 8)])
def test_rows_with_nan_on_the_mid_kernels(engine, D, K):
    from oracle import mimo_oracle as O
    from scipy.special import logsumexp
    rng = np.random.default_rng(8100 + 10 * D + K)
    N = 20011
    Z, c, b, W = _random_problem(rng, N, D, K)
    bad = rng.choice(N, size=29, replace=False)
    Zn = Z.copy()
    Zn[bad, rng.integers(0, D, size=bad.size)] = np.nan
    mask = np.ones(N); mask[bad] = 0.
    Zc = Z.copy(); Zc[bad] = 0.
    engine.upload(Zn)
    engine.tune("mid_narrow_k", 1)
    assert engine.n_bad == bad.size and engine.plan(K)["kind"] == "mid"
    L = O.canonical_eval(Zc, c, b, W)
    lse = logsumexp(L, axis=0)
    n, sx, sxx = O.packed_stats(Zc, np.exp(L - lse) * mask[None, :])
    S, sc = engine.estep(c, b, W)
    assert rel_err(S.n, n) < 1e-11 and rel_err(S.sx, sx) < 1e-11 and rel_err(S.sxx, sxx) < 1e-11
    assert abs(sc[0] - lse.sum()) < 1e-12 * max(1., abs(lse.sum()))
    assert abs(S.gating_counts.sum() - N) < 1e-9 * N
    engine.upload(Z)


@pytest.mark.parametrize("D,K", [(10, 24), (12, 40), (14, 33), (16, 48), (17, 16), (20, 32), (24, 12), (28, 40), (32, 17), (32, 48), (21, 9), (30, 5)])
def test_mid_label_pass_vs_oracle(engine, D, K):
    """The mid kernel's label mode (E-step on row-owner waves with register features, inverse-CDF draw on the unnormalised cumulative sums
    over the lane's contiguous quarter of the components) + the label-statistics kernels: labels bit-exact for host uniforms and the
    Philox stream, counts exact, statistics against the oracle, a second sweep bit-identical, rows with NaN, few workgroups."""
    from oracle import mimo_oracle as O
    rng = np.random.default_rng(9100 + 40 * D + K)
    engine.tune("mid_labels_min_d", 10)
    try:
        for N in (1, 17, 20011):
            Z, c, b, W = _random_problem(rng, N, D, K)
            engine.upload(Z)
            assert engine.plan(K, gibbs=True)["kind"] == "mid"
            if N == 20011:
                engine.tune("num_cu", 3)
            L = O.canonical_eval(Z, c, b, W)
            u = rng.random(N)
            lab, S = engine.gibbs_labels(c, b, W, u=u)
            ref = O.sample_discrete_from_log(L, u)
            assert np.array_equal(lab, ref)
            n, sx, sxx = O.packed_stats(Z, O.one_hot(ref, K))
            assert np.array_equal(S.n, n) and rel_err(S.sx, sx) < 1e-11 and rel_err(S.sxx, sxx) < 1e-11
            lab_p, Sp = engine.gibbs_labels(c, b, W, seed=3, sweep=7)
            assert np.array_equal(lab_p, O.sample_discrete_from_log(L, O.philox_uniforms(3, np.arange(N), 7)))
            lab_q, Sq = engine.gibbs_labels(c, b, W, seed=3, sweep=7)
            assert np.array_equal(lab_q, lab_p) and np.array_equal(Sq.sxx, Sp.sxx)
            engine.set_row_offset(1000)
            lab_o, _ = engine.gibbs_labels(c, b, W, seed=3, sweep=7, stats=False)
            assert np.array_equal(lab_o, O.sample_discrete_from_log(L, O.philox_uniforms(3, 1000 + np.arange(N), 7)))
            engine.set_row_offset(0)
            engine.tune("num_cu", 0)
        # rows with NaN: every label is drawn (normaliser-only log-density), the statistics leave the rows out
        bad = rng.choice(N, size=23, replace=False)
        Zn = Z.copy(); Zn[bad, 0] = np.nan
        Zc = Z.copy(); Zc[bad] = 0.
        mask = np.ones(N); mask[bad] = 0.
        engine.upload(Zn)
        Lc = O.canonical_eval(Zc, c, b, W)
        labn, G = engine.gibbs_labels(c, b, W, seed=4, sweep=2)
        refn = O.sample_discrete_from_log(Lc, O.philox_uniforms(4, np.arange(N), 2))
        gn, _, gsxx = O.packed_stats(Zc, O.one_hot(refn, K) * mask[None, :])
        assert np.array_equal(labn, refn) and np.array_equal(G.n, gn) and rel_err(G.sxx, gsxx) < 1e-11 and G.gating_counts.sum() == N
    finally:
        engine.tune("mid_labels_min_d", 0)
        engine.tune("num_cu", 0)
