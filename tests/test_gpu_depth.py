"""GPU (-m gpu): depth tests — the branches only a LONG walk reaches.  Full BASELINE sizes through size-independent properties
(configs C3, C4 and the one-pass label statistics beyond its range cap), and every kernel family with its grid cut down to a
handful of workgroups (mimo_tune "num_cu") so that each workgroup walks many tiles / steps / ranges at sizes the oracle finishes
in seconds.  (C2 at full size: test_gpu_parity.py::test_full_size_properties.)"""
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


def _tiled_rows(rng, N, D, K, scale=1e-3):
    """N rows from a 4096-row random block (+ a ramp, so that no two rows are equal) and parameters for it: full-size inputs
    without drawing N x D normals."""
    Z, c, b, W = _random_problem(rng, 4096, D, K)
    Z = np.ascontiguousarray(np.tile(Z, (N // 4096 + 1, 1))[:N] + scale * np.arange(N)[:, None] / N)
    return Z, c, b, W


def test_full_size_properties_c3_gibbs_sweep(engine):
    """BASELINE config 3 at full N (1e7 x 8, K = 256): the row-owner label kernel + the slot-table label statistics with the
    histogram counted inside the label kernel, 64-bit row offsets.  Size-independent properties: counts = bincount of the labels
    and sum to N; sum_k of the first / second moments = the column sums / Gram matrix of the data (every row carries one
    label); one component against a direct sum of its rows; a second sweep returns the same bits; the statistics of the same
    labels handed in as a vector are the same bits; halves add up; Philox labels of a row block do not depend on where the block
    starts."""
    N, D, K = 10_000_000, 8, 256
    rng = np.random.default_rng(11)
    Z, c, b, W = _tiled_rows(rng, N, D, K)
    c = c + 2. * rng.standard_normal(K)                       # uneven component weights: an uneven slot table
    engine.upload(Z)
    assert engine.plan(K, gibbs=True)["kind"] == "rowwave"
    lab, G = engine.gibbs_labels(c, b, W, seed=5, sweep=1)
    cnt = np.bincount(lab, minlength=K)
    assert lab.min() >= 0 and lab.max() < K and np.array_equal(G.n, cnt) and cnt.sum() == N
    assert rel_err(G.sx.sum(axis=0), Z.sum(axis=0)) < 1e-11 and rel_err(G.sxx.sum(axis=0), Z.T @ Z) < 1e-11
    k = int(np.argmax(cnt))
    Zk = Z[lab == k]
    assert rel_err(G.sx[k], Zk.sum(axis=0)) < 1e-11 and rel_err(G.sxx[k], Zk.T @ Zk) < 1e-11
    lab2, G2 = engine.gibbs_labels(c, b, W, seed=5, sweep=1)
    assert np.array_equal(lab2, lab) and np.array_equal(G2.sxx, G.sxx) and np.array_equal(G2.sx, G.sx)
    L = engine.label_stats(lab, K)
    assert np.array_equal(L.n, G.n) and np.array_equal(L.sxx, G.sxx) and np.array_equal(L.sx, G.sx)
    half = N // 2 + 77
    engine.upload(Z[:half]); _, Ga = engine.gibbs_labels(c, b, W, seed=5, sweep=1, return_labels=False)
    engine.upload(Z[half:]); engine.set_row_offset(half)
    lab_b, Gb = engine.gibbs_labels(c, b, W, seed=5, sweep=1)
    engine.set_row_offset(0)
    assert np.array_equal(lab_b, lab[half:])
    assert np.array_equal(Ga.n + Gb.n, G.n) and rel_err(Ga.sxx + Gb.sxx, G.sxx) < 1e-12 and rel_err(Ga.sx + Gb.sx, G.sx) < 1e-12


def test_full_size_properties_c4_ilr_softmax_pass(engine):
    """BASELINE config 4 at full N (5e6 joint rows z = [x, y] of 8 + 4 columns, K = 64): sum_k n_k = N, sum_k of the moments = the
    data's own (responsibilities sum to one), bit-identical second pass, halves add up (statistics and sum_n lse_n), Philox
    labels of the second half."""
    N, D, K = 5_000_000, 12, 64
    rng = np.random.default_rng(12)
    Z, c, b, W = _tiled_rows(rng, N, D, K)
    engine.upload(Z)
    assert engine.plan(K)["kind"] == "fused"
    S, sc = engine.estep(c, b, W)
    assert abs(S.n.sum() - N) < 1e-10 * N
    assert rel_err(S.sx.sum(axis=0), Z.sum(axis=0)) < 1e-11 and rel_err(S.sxx.sum(axis=0), Z.T @ Z) < 1e-11
    S2, sc2 = engine.estep(c, b, W)
    assert np.array_equal(S2.sxx, S.sxx) and sc2[0] == sc[0]
    lab, G = engine.gibbs_labels(c, b, W, seed=2, sweep=9)
    assert np.array_equal(G.n, np.bincount(lab, minlength=K)) and rel_err(G.sxx.sum(axis=0), Z.T @ Z) < 1e-11
    half = N // 2 - 13
    engine.upload(Z[:half]); Sa, sca = engine.estep(c, b, W)
    engine.upload(Z[half:]); engine.set_row_offset(half); Sb, scb = engine.estep(c, b, W)
    lab_b, _ = engine.gibbs_labels(c, b, W, seed=2, sweep=9, stats=False)
    engine.set_row_offset(0)
    assert rel_err(Sa.sxx + Sb.sxx, S.sxx) < 1e-12 and rel_err(Sa.n + Sb.n, S.n) < 1e-12
    assert abs((sca[0] + scb[0]) - sc[0]) < 1e-12 * abs(sc[0]) and np.array_equal(lab_b, lab[half:])


def test_full_size_properties_one_pass_label_statistics(engine):
    """label_stats_sorted_kernel at N = 1.2e7, Dz = 20, K = 96: 46 875 tiles on 512 workgroups need ranges of 92 tiles, the cap is
    80, so 74 workgroups take a SECOND range (first range writes, later ranges add).  counts = bincount, sum_k moments = the
    data's, one component against a direct sum, a second launch returns the same bits, halves add up."""
    N, D, K = 12_000_000, 20, 96
    rng = np.random.default_rng(13)
    Z, _, _, _ = _tiled_rows(rng, N, D, 1)
    p = rng.random(K) ** 3
    lab = rng.choice(K, size=N, p=p / p.sum()).astype(np.int32)
    lab[lab == 5] = 6                                          # an empty component
    engine.upload(Z)
    S = engine.label_stats(lab, K)
    cnt = np.bincount(lab, minlength=K)
    assert np.array_equal(S.n, cnt) and S.n[5] == 0 and not S.sxx[5].any()
    assert rel_err(S.sx.sum(axis=0), Z.sum(axis=0)) < 1e-11 and rel_err(S.sxx.sum(axis=0), Z.T @ Z) < 1e-11
    k = int(np.argmin(np.where(cnt > 0, cnt, N)))
    Zk = Z[lab == k]
    assert rel_err(S.sx[k], Zk.sum(axis=0)) < 1e-11 and rel_err(S.sxx[k], Zk.T @ Zk) < 1e-11
    S2 = engine.label_stats(lab, K)
    assert np.array_equal(S2.sxx, S.sxx) and np.array_equal(S2.sx, S.sx)
    half = N // 2 + 1000
    engine.upload(Z[:half]); Sa = engine.label_stats(lab[:half], K)
    engine.upload(Z[half:]); Sb = engine.label_stats(lab[half:], K)
    assert np.array_equal(Sa.n + Sb.n, S.n) and rel_err(Sa.sxx + Sb.sxx, S.sxx) < 1e-12


@pytest.mark.parametrize("ranges", [2, 3, 7])
@pytest.mark.parametrize("D,K", [(20, 96), (32, 40), (17, 256)])
def test_one_pass_label_statistics_over_several_ranges_per_workgroup(engine, D, K, ranges):
    """label_stats_sorted_kernel with 8 workgroups (mimo_tune "num_cu" = 4) and the range cap lowered (mimo_tune "sorted_range") so
    that every workgroup walks 2, 3 or 7 ranges: "first range writes, later ranges add", components that start in one range and
    go on in the next, components without a row in a workgroup's first range — against the oracle, bit-identical on a second
    launch, and through a whole Gibbs sweep (streamed label kernel in front)."""
    from oracle import mimo_oracle as O
    N = 100_003
    rng = np.random.default_rng(3300 + D + K + ranges)
    Z, c, b, W = _random_problem(rng, N, D, K)
    p = rng.random(K) ** 4
    lab = rng.choice(K, size=N, p=p / p.sum()).astype(np.int32)
    head = lab[:30_000]
    head[head % 3 == 0] = 1                                    # components that are absent from the first ranges
    engine.upload(Z)
    tiles = (N + 255) // 256
    engine.tune("num_cu", 4)
    engine.tune("sorted_range", -(-tiles // (8 * ranges)))
    try:
        S = engine.label_stats(lab, K)
        n, sx, sxx = O.packed_stats(Z, O.one_hot(lab, K))
        assert np.array_equal(S.n, n) and rel_err(S.sx, sx) < 1e-11 and rel_err(S.sxx, sxx) < 1e-11
        S2 = engine.label_stats(lab, K)
        assert np.array_equal(S2.sxx, S.sxx) and np.array_equal(S2.sx, S.sx)
        labg, G = engine.gibbs_labels(c, b, W, seed=8, sweep=2)
        L = O.canonical_eval(Z, c, b, W)
        ref = O.sample_discrete_from_log(L, O.philox_uniforms(8, np.arange(N), 2))
        assert np.array_equal(labg, ref)
        gn, gsx, gsxx = O.packed_stats(Z, O.one_hot(ref, K))
        assert np.array_equal(G.n, gn) and rel_err(G.sx, gsx) < 1e-11 and rel_err(G.sxx, gsxx) < 1e-11
    finally:
        engine.tune("num_cu", 0)
        engine.tune("sorted_range", 0)


@pytest.mark.parametrize("D,K", [(32, 210), (25, 224), (28, 200), (16, 256), (20, 72)])
def test_streamed_label_kernel_over_many_workgroup_steps(engine, D, K):
    """gibbs_stream_kernel with 8 workgroups (mimo_tune "num_cu" = 8): every workgroup takes 6 steps of 128 rows + a ragged
    seventh — the cyclic chunk walk across step boundaries in both buffer parities, incl. the 16-row-block variant
    ((32, 210), (25, 224)) — labels bit-exact for both uniform sources, statistics of the sweep, second launch identical."""
    from oracle import mimo_oracle as O
    N = 8 * 128 * 6 + 333
    rng = np.random.default_rng(5100 + 10 * D + K)
    Z, c, b, W = _random_problem(rng, N, D, K)
    engine.upload(Z)
    engine.tune("num_cu", 8)
    try:
        plan = engine.plan(K, gibbs=True)
        assert plan["kind"] == "rowwave" and plan["workgroups"] == 8
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
    finally:
        engine.tune("num_cu", 0)


@pytest.mark.parametrize("D,K,gibbs", [(16, 64, False), (8, 256, True), (8, 32, False), (2, 50, False), (2, 4, False), (32, 128, False),
                                       (12, 6, True), (20, 16, False), (8, 100, False), (12, 64, True), (3, 100, True)])
def test_few_workgroups_many_tiles_each(engine, D, K, gibbs):
    """Every kernel family with the grid cut to a handful of workgroups (mimo_tune "num_cu" = 3): each workgroup walks dozens of
    tiles / steps — the persistent loops, ring wraps and prefetch tails at a depth the full-size grid only reaches at N ~ 1e7."""
    from oracle import mimo_oracle as O
    from scipy.special import logsumexp
    N = 20011
    rng = np.random.default_rng(6100 + 10 * D + K)
    Z, c, b, W = _random_problem(rng, N, D, K)
    engine.upload(Z)
    engine.tune("num_cu", 3)
    try:
        L = O.canonical_eval(Z, c, b, W)
        if gibbs:
            lab, G = engine.gibbs_labels(c, b, W, seed=4, sweep=4)
            ref = O.sample_discrete_from_log(L, O.philox_uniforms(4, np.arange(N), 4))
            assert np.array_equal(lab, ref)
            n, sx, sxx = O.packed_stats(Z, O.one_hot(ref, K))
            assert np.array_equal(G.n, n) and rel_err(G.sx, sx) < 1e-11 and rel_err(G.sxx, sxx) < 1e-11
        else:
            lse = logsumexp(L, axis=0)
            n, sx, sxx = O.packed_stats(Z, np.exp(L - lse))
            S, sc = engine.estep(c, b, W)
            assert rel_err(S.n, n) < 1e-11 and rel_err(S.sx, sx) < 1e-11 and rel_err(S.sxx, sxx) < 1e-11
            assert abs(sc[0] - lse.sum()) < 1e-12 * abs(lse.sum())
            S2, sc2 = engine.estep(c, b, W)
            assert np.array_equal(S2.sxx, S.sxx) and sc2[0] == sc[0]
    finally:
        engine.tune("num_cu", 0)


def test_tune_rejects_unknown_keys_and_values(engine):
    with pytest.raises(ValueError):
        engine.tune("no_such_key", 1)
    with pytest.raises(ValueError):
        engine.tune("sorted_range", 81)
    with pytest.raises(ValueError):
        engine.tune("num_cu", -1)


def test_bind_finds_one_edited_element_of_a_64_mb_array(engine):
    """engine.bind() on the real engine: a 64 MB array is bound, one element is edited in place — the
    sampled fingerprint does not move —, and the next call still returns the statistics of the EDITED array: every byte is
    compared (helper thread: mimo_host_checksum; device: mimo_data_checksum from the upload's NaN scan) behind the first pass,
    the array uploaded again and the pass repeated, with a warning.  Untouched arrays: no upload, no warning; a row swap: same."""
    import warnings
    from mimo_amd import engine as E
    rng = np.random.default_rng(21)
    N, D, K = 1_000_000, 8, 5
    X = np.ascontiguousarray(rng.standard_normal((N, D)))             # 64 MB
    _, c, b, W = _random_problem(rng, 1, D, K)
    E.unbind(engine)
    E.bind(engine, X)
    assert engine.data_checksum() == E._word_checksum(X)              # host and device compute the same function
    S0, sc0 = engine.estep(c, b, W)
    n_up = engine._upload_count
    E.bind(engine, X)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        S1, sc1 = engine.estep(c, b, W)
    assert engine._upload_count == n_up and np.array_equal(S1.sxx, S0.sxx)
    X[654_321, 3] = 77.0
    assert E._bind_key(X) == engine._bound_key                        # the sample sees nothing
    E.bind(engine, X)
    with pytest.warns(RuntimeWarning, match="edited in place"):
        S2, sc2 = engine.estep(c, b, W)
    assert engine._upload_count == n_up + 1
    engine.upload(X)
    S3, sc3 = engine.estep(c, b, W)
    assert np.array_equal(S2.sxx, S3.sxx) and sc2[0] == sc3[0] and not np.array_equal(S2.sxx, S0.sxx)
    # the asynchronous form the VI driver uses, and a row swap
    E.unbind(engine); E.bind(engine, X)
    X[[10, 900_000]] = X[[900_000, 10]]
    E.bind(engine, X)
    engine.estep_async(c, b, W)
    with pytest.warns(RuntimeWarning, match="edited in place"):
        S4, sc4 = engine.estep_wait()
    lab, _ = engine.gibbs_labels(c, b, W, seed=1, sweep=1, stats=False)
    engine.upload(X)
    S5, sc5 = engine.estep(c, b, W)
    lab5, _ = engine.gibbs_labels(c, b, W, seed=1, sweep=1, stats=False)
    assert np.array_equal(S4.sxx, S5.sxx) and np.array_equal(lab, lab5)
    E.unbind(engine)


def test_plan_with_data_agrees_with_the_shape_router(engine):
    """mimo_plan (context + resident data) and mimo_plan_shape (host-only, what ROUTING.md is generated from) take the same decision."""
    import ctypes as C
    from mimo_amd import _lib
    lib = _lib.load()
    kinds = {1: "fused", 2: "two-stage", 3: "small", 4: "rowwave", 5: "rowwave-vi", 6: "narrow", 7: "mid"}
    N = 140_000
    for D in (1, 2, 3, 4, 5, 8, 9, 10, 12, 13, 16, 17, 20, 24, 27, 32):
        engine.upload(np.zeros((N, D)))
        for K in (1, 4, 8, 12, 16, 17, 32, 33, 48, 64, 65, 96, 97, 128, 129, 200, 256):
            for gibbs in (False, True):
                out = (C.c_int64 * 8)()
                assert lib.mimo_plan_shape(D, K, 0, N, 1 if gibbs else 0, out, None, 0) == 0
                p = engine.plan(K, gibbs=gibbs)
                assert p["kind"] == kinds[out[0]] and p["kernels_per_pass"] == out[1] and p["data_passes"] == out[4], (D, K, gibbs, p, list(out))


@pytest.mark.gpu
def test_graft_entry_smoke_runs():
    """__graft_entry__.smoke() — what the driver runs on a fresh box before the bench — inside the suite: its routing assertions
    (which kernel family serves which of its shapes) go stale silently otherwise, as they did when the mid kernels took Dz = 20, K = 80."""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    entry = importlib.import_module("__graft_entry__")
    entry.smoke()
