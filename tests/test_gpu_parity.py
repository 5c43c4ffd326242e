"""GPU (-m gpu): the HIP path, called through the C ABI (ctypes -> libmimo_hip.so), against the
reference's golden vectors and against the CPU oracle on seeded inputs.  Tolerances: labels and
counts bit-exact; float tables / statistics / ELBO 1e-9 relative (north star: 1e-6)."""
import numpy as np
import pytest

from conftest import GMM_CASES, ILR_CASES, GIBBS_CASES, GMM_FULLK_CASES, ILR_FULLK_CASES, rel_err, load_golden
import model_checks as mc

pytestmark = pytest.mark.gpu

GPU_GMM = list(GMM_CASES) + GMM_FULLK_CASES   # incl. Dz=32 (two-stage path) and the full-K fixtures of C3 / C5


@pytest.mark.parametrize("name", GPU_GMM)
def test_gmm_tables_stats_elbo(name, engine):
    mc.check_gmm_case(name, engine)


@pytest.mark.parametrize("name", GPU_GMM)
def test_gmm_vi_trace(name, engine):
    mc.check_gmm_vi_trace(name, engine)


@pytest.mark.parametrize("name", GIBBS_CASES)
def test_gibbs_trace(name, engine):
    mc.check_gibbs_trace(name, engine)


@pytest.mark.parametrize("name", ILR_CASES + ILR_FULLK_CASES)
def test_ilr_tables_stats_elbo(name, engine):
    mc.check_ilr_case(name, engine)


@pytest.mark.parametrize("name", ILR_CASES + ILR_FULLK_CASES)
def test_ilr_vi_trace(name, engine):
    mc.check_ilr_vi_trace(name, engine)


def test_em_map_svi_driver_traces(engine):
    mc.check_driver_traces("drivers_d3_k5_dir", engine)


def test_ilr_scaled_gibbs_then_svi(engine):
    mc.check_ilr_svi("ilr_svi_dx2_dy1_k8", engine)


def test_tied_gmm_gibbs_vi_em(engine):
    mc.check_tied_gmm("tied_gmm_d3_k5", engine)


@pytest.mark.parametrize("name", ["diag_gmm_d3_k5", "tied_diag_gmm_d4_k6"])
def test_diag_gmm_all_drivers(engine, name):
    mc.check_diag_gmm(name, engine)


@pytest.mark.parametrize("name", ["hier_gmm_d2_k4_m2", "hier_gmm_d3_k3_m3"])
def test_hierarchical_gmm_all_drivers(engine, name):
    mc.check_hier_gmm(name, engine)


@pytest.mark.parametrize("name", ["hier_ilr_dx1_dy1_k3", "hier_ilr_dx2_dy2_k4"])
def test_hierarchical_ilr_tied_activation(engine, name):
    mc.check_hier_ilr(name, engine)


@pytest.mark.parametrize("name", ["tied_ilr_sine_k8", "tied_ilr_dx3_dy2_k6"])
def test_tied_ilr_flow_and_prediction(name, engine):
    mc.check_tied_ilr_prediction(name, engine)


@pytest.mark.parametrize("N,dx,dy,K,affine", [(1000, 1, 1, 8, True), (777, 3, 2, 6, True), (4099, 8, 4, 64, True),
                                              (513, 16, 8, 50, False), (300, 32, 3, 5, True), (1, 2, 2, 3, True),
                                              (40001, 8, 4, 64, True), (2_000_003, 1, 1, 50, True), (9001, 5, 8, 7, True), (777, 8, 1, 3, False)])
def test_predict_kernel_vs_oracle(engine, N, dx, dy, K, affine):
    """mimo_predict against its canonical-level restatement: mixture / arg-max moments and nlpd."""
    from oracle import mimo_oracle as O
    rng = np.random.default_rng(N + dx + K)
    Z, c, b, W = _random_problem(rng, N, dx, K)
    dc = dx + (1 if affine else 0)
    M = rng.standard_normal((K, dy, dc))
    A = rng.standard_normal((K, dc, dc)); Q = A @ A.transpose(0, 2, 1) / dc + 0.1 * np.eye(dc)
    A = rng.standard_normal((K, dy, dy)); P = A @ A.transpose(0, 2, 1) / dy + 0.5 * np.eye(dy)
    Cc, ld = np.linalg.inv(P), np.linalg.slogdet(P)[1]
    y = rng.standard_normal((N, dy)) * 3.
    engine.upload(Z)
    for mode in ("average", "mode"):
        mu, covar, nlpd = engine.predict(c, b, W, M, Q, Cc, affine=affine, mode=mode, y=y, P=P, ld=ld)
        rmu, rcov, rnl = O.predict_canonical(Z, c, b, W, M, Q, Cc, affine, mode, y, P, ld)
        assert rel_err(mu, rmu) < 1e-11 and rel_err(covar, rcov) < 1e-10 and rel_err(nlpd, rnl) < 1e-11, mode
        mu2, covar2, none = engine.predict(c, b, W, M, Q, Cc, affine=affine, mode=mode)
        assert none is None and np.array_equal(mu2, mu) and np.array_equal(covar2, covar)
        mu3, (var3, sd3), _ = engine.predict(c, b, W, M, Q, Cc, affine=affine, mode=mode, variance='diagonal')     # MIMO_F_DIAG_VAR
        assert np.array_equal(mu3, mu) and np.array_equal(var3, np.diagonal(covar, axis1=1, axis2=2)) and np.array_equal(sd3, np.sqrt(var3))


def test_predict_with_device_resident_outputs(engine):
    """mimo_predict_flags(MIMO_F_DEVICE_IN | MIMO_F_DEVICE_OUT): inputs attached as a device tensor, y and the outputs device tensors —
    the same numbers as the host-array form, nothing but the parameter blocks crossing PCIe."""
    import torch
    from oracle import mimo_oracle as O
    rng = np.random.default_rng(77)
    N, dx, dy, K = 5003, 3, 2, 9
    Z, c, b, W = _random_problem(rng, N, dx, K)
    dc = dx + 1
    M = rng.standard_normal((K, dy, dc))
    A = rng.standard_normal((K, dc, dc)); Q = A @ A.transpose(0, 2, 1) / dc + 0.1 * np.eye(dc)
    B = rng.standard_normal((K, dy, dy)); Cc = B @ B.transpose(0, 2, 1) / dy + 0.2 * np.eye(dy)
    P = np.linalg.inv(Cc); ld = np.linalg.slogdet(P)[1]
    y = rng.standard_normal((N, dy))
    engine.upload(Z)
    mu0, cov0, nl0 = engine.predict(c, b, W, M, Q, Cc, y=y, P=P, ld=ld)
    Zd = torch.from_numpy(Z).cuda()
    engine.upload(Zd)
    yd = torch.from_numpy(y).cuda()
    mu = torch.empty((N, dy), dtype=torch.float64, device="cuda"); cov = torch.empty((N, dy, dy), dtype=torch.float64, device="cuda")
    nl = torch.empty(N, dtype=torch.float64, device="cuda")
    engine.predict_device(c, b, W, M, Q, Cc, mu.data_ptr(), cov.data_ptr(), y_ptr=yd.data_ptr(), P=P, ld=ld, nlpd_ptr=nl.data_ptr())
    engine.estep(c, b, W, stats=False)          # (a synchronous call on the same stream: the prediction before it is done)
    assert np.array_equal(mu.cpu().numpy(), mu0) and np.array_equal(cov.cpu().numpy(), cov0) and np.array_equal(nl.cpu().numpy(), nl0)
    engine.upload(Z)


def test_predict_rejects_bad_shapes(engine):
    rng = np.random.default_rng(0)
    Z, c, b, W = _random_problem(rng, 64, 2, 3)
    engine.upload(Z)
    M, Q = rng.standard_normal((3, 9, 3)), np.tile(np.eye(3), (3, 1, 1))
    with pytest.raises(Exception):
        engine.predict(c, b, W, M, Q, np.tile(np.eye(9), (3, 1, 1)))          # dy = 9 > 8: unsupported
    with pytest.raises(ValueError):
        engine.predict(c, b, W, M[:, :2], Q[:, :2, :2], np.tile(np.eye(2), (3, 1, 1)))   # dc mismatch


def test_unsupported_shapes_fail_loudly(engine):
    from mimo_amd import _lib
    with pytest.raises(_lib.MimoHipError):
        engine.upload(np.zeros((10, 40)))
    engine.upload(np.zeros((10, 4)))
    with pytest.raises(_lib.MimoHipError):
        engine.estep(np.zeros(300), np.zeros((300, 4)), np.tile(np.eye(4), (300, 1, 1)))
    with pytest.raises(ValueError):
        engine.label_stats(np.array([0, 5] * 5), 3)


def _random_problem(rng, N, D, K):
    Z = rng.standard_normal((N, D)) * 2.0 + rng.standard_normal(D)
    A = rng.standard_normal((K, D, D))
    W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D)
    mu = rng.standard_normal((K, D)) * 2
    b = np.einsum('kde,ke->kd', W, mu)
    c = -0.5 * np.einsum('kd,kd->k', mu, b) + rng.standard_normal(K) * 0.1
    return Z, c, b, W


@pytest.mark.parametrize("N,D,K", [(0, 3, 2), (1, 1, 1), (31, 16, 16), (32, 2, 4), (33, 5, 70), (4099, 8, 256),
                                    (4099, 12, 64), (20011, 16, 64), (1000, 9, 200), (777, 13, 17),
                                    # two-stage path (chunked E-step + column-group statistics)
                                    (2051, 32, 128), (515, 32, 16), (700, 12, 200), (333, 20, 256), (100, 17, 3),
                                    # 8-wave statistics kernel of the two-stage path (3 <= K16 <= 8): every blocks-per-wave instantiation
                                    (3000, 24, 64), (1500, 17, 40), (999, 28, 100), (640, 20, 72), (4097, 32, 90),
                                    # ... and the same kernels at 10 <= Dz <= 16 with K > 64
                                    (3000, 16, 128), (2500, 14, 100), (1200, 16, 200), (900, 13, 256), (1000, 12, 128), (801, 10, 90),
                                    (2222, 28, 40), (1111, 15, 35),
                                    # ... ragged and tiny row counts through the same kernels
                                    (1, 32, 128), (31, 20, 72), (33, 24, 100), (65, 16, 200), (0, 32, 128),
                                    (700, 32, 32), (650, 20, 24), (900, 17, 30),
                                    # one feature chunk per tile (Dz = 10 .. 12, K > 64) in the pipelined E-step
                                    (3001, 12, 200), (2000, 10, 256), (1500, 11, 100), (33, 12, 129),
                                    # 128 < K <= 192: three row blocks per wave (E-step; statistics where that saves a launch)
                                    (1300, 32, 192), (1100, 24, 144), (900, 16, 180), (700, 8, 100),
                                    (4000, 8, 192), (2500, 5, 150), (3100, 9, 129), (33, 7, 177)])
def test_engine_vs_oracle_seeded(engine, N, D, K):
    """Every entry point against the oracle's direct evaluation, ragged / empty / maximal shapes."""
    from oracle import mimo_oracle as O
    from scipy.special import logsumexp
    rng = np.random.default_rng(1000 * D + K)
    Z, c, b, W = _random_problem(rng, N, D, K)
    engine.upload(Z)
    L = O.canonical_eval(Z, c, b, W)
    lse = logsumexp(L, axis=0) if N else np.zeros(0)
    R = np.exp(L - lse)
    S, sc = engine.estep(c, b, W, keep_resp=True, keep_logp=True, keep_lse=True)
    n, sx, sxx = O.packed_stats(Z, R)
    assert rel_err(engine.get_logp(), L) < 1e-12 and rel_err(engine.get_lse(), lse) < 1e-12
    assert rel_err(engine.get_resp(), R) < 1e-11
    assert rel_err(S.n, n) < 1e-11 and rel_err(S.sx, sx) < 1e-11 and rel_err(S.sxx, sxx) < 1e-11
    if N:
        assert abs(sc[0] - lse.sum()) < 1e-12 * abs(lse.sum())
        assert abs(sc[2] + np.nansum(R * np.log(np.where(R > 0, R, 1)))) < 1e-9 * max(1.0, abs(sc[2]))
        assert abs(engine.table_entropy() - sc[2]) < 1e-9 * max(1.0, abs(sc[2]))
    Wt = rng.random((K, N))
    S2 = engine.weighted_stats(Wt)
    n2, sx2, sxx2 = O.packed_stats(Z, Wt)
    assert rel_err(S2.n, n2) < 1e-11 and rel_err(S2.sxx, sxx2) < 1e-11
    u = rng.random(N)
    lab, S3 = engine.gibbs_labels(c, b, W, u=u)
    ref = O.sample_discrete_from_log(L, u) if N else np.zeros(0, np.int32)
    assert np.array_equal(lab, ref)
    assert np.array_equal(S3.n, np.bincount(ref, minlength=K))
    lab_p, _ = engine.gibbs_labels(c, b, W, seed=99, sweep=7, stats=False)
    ref_p = O.sample_discrete_from_log(L, O.philox_uniforms(99, np.arange(N), 7)) if N else ref
    assert np.array_equal(lab_p, ref_p)
    assert np.array_equal(engine.get_labels(), ref_p)


@pytest.mark.parametrize("D,K", [(7, 20), (8, 33), (10, 64), (11, 5), (13, 40), (15, 64), (16, 64), (8, 200), (3, 256),
                                 (32, 128), (20, 16), (12, 100), (16, 128), (24, 70),
                                 (16, 16), (16, 3), (12, 16), (9, 4), (7, 1), (14, 11),      # K <= 32, Dz >= 7: split kernels
                                 (16, 32), (16, 17), (11, 25), (8, 32), (7, 30)])
def test_many_tiles_per_workgroup(engine, D, K):
    """N large enough that every workgroup walks several tiles (the Theta ring wraps from tile to tile):
    fused E-step, Gibbs labels and statistics against the oracle for every ring geometry."""
    from oracle import mimo_oracle as O
    from scipy.special import logsumexp
    N = 32 * 512 * 3 + 77
    rng = np.random.default_rng(100 * D + K)
    Z, c, b, W = _random_problem(rng, N, D, K)
    engine.upload(Z)
    L = O.canonical_eval(Z, c, b, W)
    lse = logsumexp(L, axis=0)
    n, sx, sxx = O.packed_stats(Z, np.exp(L - lse))
    S, sc = engine.estep(c, b, W)                       # fast mode
    assert rel_err(S.n, n) < 1e-11 and rel_err(S.sx, sx) < 1e-11 and rel_err(S.sxx, sxx) < 1e-11
    assert abs(sc[0] - lse.sum()) < 1e-12 * abs(lse.sum())
    S2, sc2 = engine.estep(c, b, W, keep_lse=True)      # generic mode
    assert rel_err(S2.sxx, sxx) < 1e-11 and rel_err(engine.get_lse(), lse) < 1e-12
    lab, S3 = engine.gibbs_labels(c, b, W, seed=3, sweep=9)
    ref = O.sample_discrete_from_log(L, O.philox_uniforms(3, np.arange(N), 9))
    assert np.array_equal(lab, ref) and np.array_equal(S3.n, np.bincount(ref, minlength=K))


@pytest.mark.parametrize("D,K", [(15, 64), (8, 256), (3, 256), (16, 64), (16, 16), (10, 7), (16, 32), (9, 20)])
def test_repeated_launches_are_bit_identical(engine, D, K):
    """Every launch of every mode returns the SAME bits (fixed summation order, no float atomics) with all
    workgroups co-resident and several tiles per workgroup.  Regression for a lane-layout experiment whose
    kernels were right on the first tile of a workgroup and sporadically wrong on later ones when two workgroups
    shared a CU (tools/stress_generic.py is the long form of this test)."""
    from oracle import mimo_oracle as O
    from scipy.special import logsumexp
    N = 32 * 512 * 3 + 77
    rng = np.random.default_rng(100 * D + K)
    Z, c, b, W = _random_problem(rng, N, D, K)
    engine.upload(Z)
    L = O.canonical_eval(Z, c, b, W)
    _, _, sxx = O.packed_stats(Z, np.exp(L - logsumexp(L, axis=0)))
    ref = O.sample_discrete_from_log(L, O.philox_uniforms(5, np.arange(N), 1))
    for kw in ({}, dict(keep_lse=True), dict(entropy_split=True)):
        first = None
        for _ in range(6):
            S, sc = engine.estep(c, b, W, **kw)
            if first is None:
                first = (S.sxx.copy(), sc[0])
                assert rel_err(S.sxx, sxx) < 1e-11
            assert np.array_equal(S.sxx, first[0]) and sc[0] == first[1]
    for _ in range(6):
        lab, S = engine.gibbs_labels(c, b, W, seed=5, sweep=1)
        assert np.array_equal(lab, ref) and np.array_equal(S.n, np.bincount(ref, minlength=K))


@pytest.mark.parametrize("D,K,N", [(2, 4, 1000), (16, 64, 40000), (8, 200, 33000), (5, 7, 0), (20, 16, 5000),
                                   (12, 9, 21000), (16, 16, 40000), (13, 27, 30000)])
def test_row_weighted_estep(engine, D, K, N):
    """mimo_estep_weighted: statistics of r_kn w_n, scalars / tables of the unweighted r_kn (hgmm.py:199-207) —
    single-pass shapes in-kernel, two-stage shapes (Dz = 20) through the table route."""
    from oracle import mimo_oracle as O
    from scipy.special import logsumexp
    rng = np.random.default_rng(1000 * D + K)
    Z, c, b, W = _random_problem(rng, max(N, 1), D, K)
    Z = Z[:N]
    w = rng.uniform(0.0, 2.0, size=N)
    engine.upload(Z)
    L = O.canonical_eval(Z, c, b, W)
    lse = logsumexp(L, axis=0) if N else np.zeros(0)
    R = np.exp(L - lse)
    n, sx, sxx = O.packed_stats(Z, R * w[None, :])
    S, sc = engine.estep(c, b, W, row_weights=w, keep_resp=True)
    scale = max(np.abs(sxx).max(), 1e-300) if N else 1.0
    assert np.abs(S.n - n).max() <= 1e-11 * max(n.max(), 1.0) and np.abs(S.sxx - sxx).max() <= 1e-11 * scale
    assert abs(sc[0] - lse.sum()) <= 1e-12 * max(abs(lse.sum()), 1.0)
    if N:
        assert rel_err(engine.get_resp(K), R) < 1e-11        # the table stays unweighted
        S1, _ = engine.estep(c, b, W, row_weights=np.ones(N))
        S0, _ = engine.estep(c, b, W)
        assert rel_err(S1.sxx, S0.sxx) < 1e-13


@pytest.mark.parametrize("D,K,N", [(2, 4, 1000), (16, 64, 40000), (32, 64, 20000), (8, 256, 33000), (32, 128, 9000),
                                   (5, 7, 0), (24, 200, 5000),
                                   (16, 16, 30000), (32, 24, 20000), (24, 8, 9000), (31, 32, 33000),    # K <= 32: split kernels
                                   (6, 40, 30011), (7, 64, 20005), (5, 33, 9001), (12, 8, 30000)])       # <= 16 / 17 .. features: narrow kernels
def test_diagonal_structure(engine, D, K, N):
    """mimo_set_structure(MIMO_STRUCT_DIAG): the 2 Dz + 1 feature kernels against the oracle with W = diag —
    tables, statistics (zero off-diagonal second moments), bound, labels; full W is rejected; switching back."""
    from oracle import mimo_oracle as O
    from scipy.special import logsumexp
    rng = np.random.default_rng(77 * D + K)
    Z, c, b, Wf = _random_problem(rng, max(N, 1), D, K)
    Z = Z[:N]
    W = Wf * np.eye(D)[None, :, :]
    L = O.canonical_eval(Z, c, b, W)
    lse = logsumexp(L, axis=0) if N else np.zeros(0)
    R = np.exp(L - lse)
    n, sx, sxx = O.packed_stats(Z, R)
    dsxx = sxx * np.eye(D)[None, :, :]
    try:
        engine.set_structure('diag')
        engine.upload(Z)
        S, sc = engine.estep(c, b, W)
        scale = max(np.abs(sxx).max(), 1e-300) if N else 1.0
        assert np.abs(S.n - n).max() <= 1e-11 * max(n.max(), 1.0) and np.abs(S.sx - sx).max() <= 1e-11 * scale
        assert np.abs(S.sxx - dsxx).max() <= 1e-11 * scale
        assert abs(sc[0] - lse.sum()) <= 1e-12 * max(abs(lse.sum()), 1.0)
        if N:
            S2, _ = engine.estep(c, b, W, keep_logp=True, keep_resp=True)
            assert rel_err(engine.get_logp(K), L) < 1e-12 and rel_err(engine.get_resp(K), R) < 1e-10
            assert np.abs(S2.sxx - dsxx).max() <= 1e-11 * scale
            lab, S3 = engine.gibbs_labels(c, b, W, seed=3, sweep=9)
            ref = O.sample_discrete_from_log(L, O.philox_uniforms(3, np.arange(N), 9))
            assert np.array_equal(lab, ref) and np.array_equal(S3.n, np.bincount(ref, minlength=K))
            Sw = engine.weighted_stats(R)
            assert np.abs(Sw.sxx - dsxx).max() <= 1e-11 * scale
            with pytest.raises((RuntimeError, ValueError)):
                engine.estep(c, b, Wf)                       # off-diagonal entries under the diagonal structure
    finally:
        engine.set_structure('full')
    if N:
        S4, _ = engine.estep(c, b, W)                        # same W through the full feature map
        assert np.abs(S4.sxx - sxx).max() <= 1e-11 * scale


@pytest.mark.parametrize("D,K,N", [(2, 4, 1000), (16, 64, 40000), (15, 64, 20000), (8, 256, 33000), (32, 128, 9000),
                                   (5, 7, 0),
                                   (8, 33, 30011), (12, 20, 20005), (6, 64, 25003), (16, 48, 9001), (11, 50, 70001)])   # narrow kernels on the linear map
def test_linear_structure_for_tied_blocks(engine, D, K, N):
    """mimo_set_structure(MIMO_STRUCT_LINEAR): one W for all components — the Dz + 1 feature kernels plus the data
    constants of the shared quadratic term reproduce the full-structure results: n_k, sum r z, the POOLED second
    moment, the bound, responsibilities and labels; tables with the quadratic term fall back to the full map."""
    from oracle import mimo_oracle as O
    from scipy.special import logsumexp
    rng = np.random.default_rng(31 * D + K)
    Z, c, b, Wf = _random_problem(rng, max(N, 1), D, K)
    Z = Z[:N]
    W = np.ascontiguousarray(np.broadcast_to(Wf[:1], Wf.shape))
    L = O.canonical_eval(Z, c, b, W)
    lse = logsumexp(L, axis=0) if N else np.zeros(0)
    R = np.exp(L - lse)
    n, sx, sxx = O.packed_stats(Z, R)
    XX = Z.T @ Z
    w = rng.uniform(0., 2., size=N)
    try:
        engine.set_structure('linear')
        engine.upload(Z)
        S, sc = engine.estep(c, b, W)
        scale = max(np.abs(XX).max(), 1e-300) if N else 1.0
        assert S.sxx is None and np.abs(S.sxx_total - XX).max() <= 1e-11 * scale
        assert np.abs(S.n - n).max() <= 1e-11 * max(n.max(), 1.0) and np.abs(S.sx - sx).max() <= 1e-11 * scale
        assert abs(sc[0] - lse.sum()) <= 1e-11 * max(abs(lse.sum()), 1.0)
        if N:
            engine.estep_async(c, b, W)
            Sa, sca = engine.estep_wait()
            assert np.array_equal(Sa.sx, S.sx) and sca[0] == sc[0]
            Sw, scw = engine.estep(c, b, W, row_weights=w, keep_resp=True)
            assert rel_err(Sw.sxx_total, (Z * w[:, None]).T @ Z) < 1e-11 and abs(scw[0] - sc[0]) <= 1e-12 * abs(sc[0])
            assert rel_err(Sw.sx, O.packed_stats(Z, R * w[None, :])[1]) < 1e-10
            assert rel_err(engine.get_resp(K), R) < 1e-9
            engine.estep(c, b, W, stats=False, keep_logp=True, keep_lse=True)          # full map behind the scenes
            assert rel_err(engine.get_logp(K), L) < 1e-12 and rel_err(engine.get_lse(), lse) < 1e-12
            lab, S3 = engine.gibbs_labels(c, b, W, seed=3, sweep=9)
            ref = O.sample_discrete_from_log(L, O.philox_uniforms(3, np.arange(N), 9))
            assert np.mean(lab != ref) < 1e-4                  # the common term only moves last-bit ties
            assert np.array_equal(S3.n, np.bincount(lab, minlength=K)) and rel_err(S3.sxx_total, XX) < 1e-11
            S4, sc4 = engine.estep(c, b, Wf)                   # not a tied block: silently the full map
            assert S4.sxx is not None
    finally:
        engine.set_structure('full')


@pytest.mark.parametrize("native", ["1", "0"])
def test_sharded_device_route_over_one_rank_rccl(engine, native, monkeypatch):
    """ShardedEngine on the GPU, both routes of the all-reduce: native = the library's own RCCL communicator (the
    collective is enqueued by libmimo_hip.so behind its kernels), otherwise kernel -> device buffer -> torch.distributed
    all_reduce -> pinned host memory.  With a one-rank group the result must equal the plain engine's, for the full and
    the 'linear' structure, synchronous, asynchronous and for the Gibbs step (the two-rank arithmetic is covered on CPU in
    test_sharded_gloo)."""
    monkeypatch.setenv("MIMO_SHARDED_NATIVE", native)
    import torch
    import torch.distributed as dist
    from mimo_amd.sharded import ShardedEngine
    from mimo_amd.engine import HipEngine
    rng = np.random.default_rng(77)
    N, D, K = 50000, 16, 64
    Z, c, b, Wf = _random_problem(rng, N, D, K)
    Wt = np.ascontiguousarray(np.broadcast_to(Wf[:1], Wf.shape))
    Z8, c8, b8, W8 = _random_problem(rng, 30011, 8, 32)           # a shape of the row-owner kernels (softmax and label pass)
    engine.upload(Z)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1,
                                device_id=torch.device("cuda:0"))
    try:
        sh = ShardedEngine(HipEngine(0), row_offset=0)
        assert sh._native == (native == "1") and sh._device_path == (native == "0")
        sh.upload(Z)
        for structure, W in (("full", Wf), ("linear", Wt), ("linear", Wf)):
            try:
                engine.set_structure(structure)
                sh.set_structure(structure)
                S0, sc0 = engine.estep(c, b, W)
                S1, sc1 = sh.estep(c, b, W)
                sh.estep_async(c, b, W)
                S2, sc2 = sh.estep_wait()
                for S, sc in ((S1, sc1), (S2, sc2)):
                    assert np.array_equal(S.n, S0.n) and np.array_equal(S.sx, S0.sx) and sc[0] == sc0[0]
                    if S0.sxx is None:
                        assert S.sxx is None and np.array_equal(S.sxx_total, S0.sxx_total)
                    else:
                        assert np.array_equal(S.sxx, S0.sxx)
                _, G0 = engine.gibbs_labels(c, b, W, seed=5, sweep=2, return_labels=False)
                _, G1 = sh.gibbs_labels(c, b, W, seed=5, sweep=2, return_labels=False)
                assert np.array_equal(G0.n, G1.n) and np.array_equal(G0.sx, G1.sx)
                if G0.sxx is None:
                    assert G1.sxx is None and np.array_equal(G1.sxx_total, G0.sxx_total)
                else:
                    assert np.array_equal(G0.sxx, G1.sxx)
            finally:
                engine.set_structure('full')
                sh.set_structure('full')
        engine.upload(Z8); sh.upload(Z8)
        assert engine.plan(32)["kind"] == "rowwave-vi" and engine.plan(32, gibbs=True)["kind"] == "rowwave"
        S0, sc0 = engine.estep(c8, b8, W8)
        S1, sc1 = sh.estep(c8, b8, W8)
        sh.estep_async(c8, b8, W8)
        S2, sc2 = sh.estep_wait()
        for S, sc in ((S1, sc1), (S2, sc2)):
            assert np.array_equal(S.n, S0.n) and np.array_equal(S.sxx, S0.sxx) and sc[0] == sc0[0]
        _, G0 = engine.gibbs_labels(c8, b8, W8, seed=5, sweep=2, return_labels=False)
        _, G1 = sh.gibbs_labels(c8, b8, W8, seed=5, sweep=2, return_labels=False)
        assert np.array_equal(G0.n, G1.n) and np.array_equal(G0.sxx, G1.sxx)
        sh.inner.close()
    finally:
        if created:
            dist.destroy_process_group()


def test_switched_off_component_and_far_clusters(engine):
    """A component with log-weight -inf (gating probability exactly 0: np.log(probs) in the host mirror) and clusters
    so far apart that l - max < -707 for most components (the clamp of the kernel's exp): responsibilities of the
    switched-off / far components are 0 to 1e-300, statistics and the bound match the oracle, no label falls on the
    switched-off component, the draw is the oracle's; ragged N so that rows past N sit in the last tile.  NaN / +inf
    parameters are rejected."""
    from oracle import mimo_oracle as O
    from scipy.special import logsumexp
    rng = np.random.default_rng(123)
    N, D, K = 3001, 4, 37
    centres = 400. * rng.standard_normal((K, D))
    lab = rng.integers(0, K, N)
    Z = centres[lab] + rng.standard_normal((N, D))
    W = np.stack(K * [np.eye(D)]) * rng.uniform(0.5, 2., K)[:, None, None]
    b = np.einsum('kde,ke->kd', W, centres)
    c = -0.5 * np.einsum('kd,kd->k', centres, b) + rng.standard_normal(K)
    c[5] = -np.inf
    with np.errstate(invalid='ignore'):
        L = O.canonical_eval(Z, c, b, W)
    L[5] = -np.inf
    lse = logsumexp(L, axis=0)
    R = np.exp(L - lse)
    n, sx, sxx = O.packed_stats(Z, R)
    engine.upload(Z)
    S, sc = engine.estep(c, b, W, keep_resp=True)
    assert rel_err(S.n, n) < 1e-11 and rel_err(S.sx, sx) < 1e-11 and rel_err(S.sxx, sxx) < 1e-11
    assert abs(sc[0] - lse.sum()) < 1e-11 * abs(lse.sum()) and np.all(np.isfinite(S.sxx))
    Rg = engine.get_resp(K)
    assert np.abs(Rg - R).max() < 1e-14 and Rg[5].max() < 1e-300 and S.n[5] < 1e-290
    # entropy split with the switched-off component: sum r l counts 0 for it (it used to add exp(-707) * -1e300 =
    # -8e-8 per datum), here with few far clusters so that the terms are not swamped, and on the chunked normalise (K > 64)
    with np.errstate(invalid='ignore'):
        srl = float(np.nansum(np.where(R > 0, R * L, 0.)))
    _, sc2 = engine.estep(c, b, W, entropy_split=True)
    assert abs(sc2[1] - srl) < 1e-11 * abs(srl) and abs(sc2[2] - (lse.sum() - srl)) < 1e-9 * max(1., abs(lse.sum() - srl))
    K2 = 70
    c3, b3, W3 = (np.concatenate([v, v[:K2 - K]]) for v in (c, b, W))
    with np.errstate(invalid='ignore'):
        L3 = O.canonical_eval(Z, c3, b3, W3)
    L3[5] = -np.inf
    lse3 = logsumexp(L3, axis=0)
    # (rows of the switched-off cluster are far from everything: |l| ~ 3e5, where exp(l - lse) loses 11 digits of the
    # weights to the rounding of lse; the reference weights are formed from l - max like the kernel's)
    E3 = np.exp(L3 - L3.max(axis=0))
    R3 = E3 / E3.sum(axis=0)
    srl3 = float(np.nansum(np.where(R3 > 0, R3 * L3, 0.)))
    _, sc3 = engine.estep(c3, b3, W3, entropy_split=True)
    assert abs(sc3[0] - lse3.sum()) < 1e-11 * abs(lse3.sum()) and abs(sc3[1] - srl3) < 1e-11 * abs(srl3)
    labels, Sg = engine.gibbs_labels(c, b, W, seed=9, sweep=4)
    ref = O.sample_discrete_from_log(L, O.philox_uniforms(9, np.arange(N), 4))
    assert np.array_equal(labels, ref) and not np.any(labels == 5)
    assert np.array_equal(Sg.n, np.bincount(ref, minlength=K))
    for bad in (np.nan, np.inf):
        c2 = c.copy(); c2[3] = bad
        with pytest.raises(ValueError):
            engine.estep(c2, b, W)
    b2 = b.copy(); b2[7, 1] = np.nan
    with pytest.raises(ValueError):
        engine.estep(c, b2, W)


def test_example_scripts_run_on_the_gpu():
    """examples/: the command-line counterparts of the reference's toy GMM / DP-GMM / sine-regression scripts run
    end to end on the HIP engine (own processes, one at a time) and recover what they should."""
    import os, re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(*args):
        r = subprocess.run([sys.executable] + list(args), cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        return r.stdout

    # per datum: the true mixture -3.40; a local mode with two clusters under one component (K = 4 Gibbs finds one in
    # about one run of seven, like the reference) -4.0 .. -4.1; one Gaussian for everything -5.6
    for method in ("gibbs", "vi", "em"):
        out = run("examples/gmm_toy.py", "--method", method, "--rows", "5000")
        assert float(re.search(r"log-likelihood per datum: (-?[0-9.]+)", out).group(1)) > -4.3, out
    out = run("examples/dpgmm_gibbs.py", "--rows", "200000", "--dim", "4", "--kmax", "64", "--clusters", "6", "--sweeps", "40")
    assert "evaluations/s" in out
    out = run("examples/ilr_sine.py", "--rows", "5000", "--experts", "12", "--iters", "60")
    assert float(re.search(r"RMSE against the noiseless curve: ([0-9.]+)", out).group(1)) < 0.5, out
    # the reference's default truncation (examples/ilr/evaluate_sine.py:35: 50 experts over dx = dy = 1): the narrow kernels
    out = run("examples/ilr_sine.py", "--rows", "5000", "--experts", "50", "--iters", "40")
    assert "softmax pass 'narrow', label pass 'narrow'" in out, out
    assert float(re.search(r"RMSE against the noiseless curve: ([0-9.]+)", out).group(1)) < 0.5, out


def test_full_size_properties(engine):
    """BASELINE config 2 shape at full N (1e7 x 16, K=64): size-independent properties —
    (i) responsibilities sum to one => sum_k n_k = N exactly to rounding; (ii) linearity: the
    statistics of the two halves add up to the statistics of the whole; (iii) run-to-run bit
    reproducibility; (iv) Philox labels of a row block do not depend on where the block starts."""
    N, D, K = 10_000_000, 16, 64
    rng = np.random.default_rng(7)
    Z, c, b, W = _random_problem(rng, 4096, D, K)
    Z = np.ascontiguousarray(np.tile(Z, (N // 4096 + 1, 1))[:N] + 1e-3 * np.arange(N)[:, None] / N)
    engine.upload(Z)
    S, sc = engine.estep(c, b, W)
    assert abs(S.n.sum() - N) < 1e-9 * N
    S_again, sc_again = engine.estep(c, b, W)
    assert np.array_equal(S.sxx, S_again.sxx) and sc[0] == sc_again[0]
    lab_full, Sg = engine.gibbs_labels(c, b, W, seed=5, sweep=1)
    assert Sg.n.sum() == N and np.array_equal(Sg.n, np.bincount(lab_full, minlength=K))
    half = N // 2
    engine.upload(Z[:half]); Sa, sca = engine.estep(c, b, W)
    engine.upload(Z[half:]); engine.set_row_offset(half); Sb, scb = engine.estep(c, b, W)
    lab_b, _ = engine.gibbs_labels(c, b, W, seed=5, sweep=1, stats=False)
    engine.set_row_offset(0)
    assert rel_err(Sa.sxx + Sb.sxx, S.sxx) < 1e-12 and rel_err(Sa.n + Sb.n, S.n) < 1e-12
    assert abs((sca[0] + scb[0]) - sc[0]) < 1e-12 * abs(sc[0])
    assert np.array_equal(lab_b, lab_full[half:])


def test_plain_c_client_of_the_abi(tmp_path):
    """The boundary is a C ABI: a C99 program (tests/abi_smoke.c, gcc, no C++ / Python in the loop) uploads a
    problem, runs the fused pass and the Philox label step through include/mimo_hip.h and must reproduce the oracle."""
    import os
    import shutil
    import subprocess
    from oracle import mimo_oracle as O
    from scipy.special import logsumexp
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "abi_smoke")
    build = subprocess.run([gcc, "-std=c99", "-O1", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "abi_smoke.c"),
                            "-L", os.path.join(root, "mimo_amd"), "-lmimo_hip", "-Wl,-rpath," + os.path.join(root, "mimo_amd"),
                            "-o", exe], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    N, D, K = 333, 5, 7
    rng = np.random.default_rng(5)
    Z, c, b, W = _random_problem(rng, N, D, K)
    text = f"{N} {D} {K}\n" + "\n".join(repr(float(v)) for a in (Z, c, b, W) for v in np.ravel(a)) + "\n"
    run = subprocess.run([exe], input=text, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "done" in run.stdout, run.stdout[-500:] + run.stderr[-2000:]
    lines = run.stdout.splitlines()
    S = np.array([float(l.split()[1]) for l in lines if l.startswith("S ")]).reshape(K, 1 + D + D * D)
    labels = np.array([int(l.split()[1]) for l in lines if l.startswith("L ")])
    sc0 = float([l for l in lines if l.startswith("estep_scalar0")][0].split()[1])
    L = O.canonical_eval(Z, c, b, W)
    lse = logsumexp(L, axis=0)
    n, sx, sxx = O.packed_stats(Z, np.exp(L - lse))
    assert rel_err(S[:, 0], n) < 1e-11 and rel_err(S[:, 1:1 + D], sx) < 1e-11
    assert rel_err(S[:, 1 + D:].reshape(K, D, D), sxx) < 1e-11 and abs(sc0 - lse.sum()) < 1e-12 * abs(lse.sum())
    assert np.array_equal(labels, O.sample_discrete_from_log(L, O.philox_uniforms(42, np.arange(N), 3)))
    assert any(l.startswith("error_message") and "K must be" in l for l in lines)


SMALL_SHAPES = [(1, 1), (1, 3), (1, 6), (1, 13), (1, 32), (2, 1), (2, 4), (2, 6), (2, 8), (2, 16), (2, 25), (2, 32),
                (3, 3), (3, 4), (3, 5), (3, 8), (3, 13), (3, 16), (4, 2), (4, 3), (4, 7), (4, 9), (4, 16)]


@pytest.mark.parametrize("D,K", SMALL_SHAPES)
@pytest.mark.parametrize("N", [1, 63, 5003, 3 * 1024 * 256 + 5])
def test_small_shape_kernel_vs_oracle(engine, D, K, N):
    """The VALU kernel for Dz <= 4, K <= 32 (mimo_small.hip) — every (Dz, lanes-per-row) geometry, ragged N down to one
    row and N large enough for several chunks per wave: tables, statistics, scalars, both label sources, table-driven and
    label-driven statistics, per-row weights, and run-to-run bit reproducibility, against the oracle."""
    from oracle import mimo_oracle as O
    from scipy.special import logsumexp
    assert engine.plan is not None
    rng = np.random.default_rng(7000 + 100 * D + K)
    Z, c, b, W = _random_problem(rng, N, D, K)
    engine.upload(Z)
    # (plain passes of the upper K range of the small-shape kernel went to the narrow kernels in round 3; its generic requests stay)
    assert engine.plan(K)["kind"] == ("narrow" if K >= (17 if D <= 2 else 9 if D == 3 else 12) else "small")
    L = O.canonical_eval(Z, c, b, W)
    lse = logsumexp(L, axis=0)
    R = np.exp(L - lse)
    n, sx, sxx = O.packed_stats(Z, R)
    tol = 1e-11
    S, sc = engine.estep(c, b, W)                                        # fast mode
    assert rel_err(S.n, n) < tol and rel_err(S.sx, sx) < tol and rel_err(S.sxx, sxx) < tol
    assert abs(sc[0] - lse.sum()) < 1e-12 * max(1., abs(lse.sum()))
    S1, sc1 = engine.estep(c, b, W)
    assert np.array_equal(S1.sxx, S.sxx) and np.array_equal(S1.n, S.n) and sc1[0] == sc[0]
    big = N > 100000
    if not big:
        Sg, scg = engine.estep(c, b, W, keep_resp=True, keep_logp=True, keep_lse=True)     # generic mode
        assert rel_err(engine.get_logp(), L) < 1e-12 and rel_err(engine.get_lse(), lse) < 1e-12
        assert rel_err(engine.get_resp(), R) < tol and rel_err(Sg.sxx, sxx) < tol
        srl = float(np.sum(R * L))
        assert abs(scg[0] - lse.sum()) < 1e-12 * max(1., abs(lse.sum())) and abs(scg[1] - srl) < 1e-11 * max(1., abs(srl))
        Wt = rng.random((K, N))
        S2 = engine.weighted_stats(Wt)
        n2, sx2, sxx2 = O.packed_stats(Z, Wt)
        assert rel_err(S2.n, n2) < tol and rel_err(S2.sx, sx2) < tol and rel_err(S2.sxx, sxx2) < tol
        w = rng.random(N) + 0.1
        Sw, scw = engine.estep(c, b, W, row_weights=w)
        nw, sxw, sxxw = O.packed_stats(Z, R * w)
        assert rel_err(Sw.n, nw) < tol and rel_err(Sw.sxx, sxxw) < tol and abs(scw[0] - lse.sum()) < 1e-12 * max(1., abs(lse.sum()))
        u = rng.random(N)
        lab, S3 = engine.gibbs_labels(c, b, W, u=u)
        ref = O.sample_discrete_from_log(L, u)
        assert np.array_equal(lab, ref) and np.array_equal(S3.n, np.bincount(ref, minlength=K))
        S4 = engine.label_stats(ref, K)
        n4, sx4, sxx4 = O.packed_stats(Z, O.one_hot(ref, K))
        assert np.array_equal(S4.n, n4) and rel_err(S4.sx, sx4) < tol and rel_err(S4.sxx, sxx4) < tol
        assert rel_err(S3.sxx, sxx4) < tol
    lab_p, Sp = engine.gibbs_labels(c, b, W, seed=99, sweep=7)
    ref_p = O.sample_discrete_from_log(L, O.philox_uniforms(99, np.arange(N), 7)) if not big else None
    if ref_p is not None:
        assert np.array_equal(lab_p, ref_p) and np.array_equal(Sp.n, np.bincount(ref_p, minlength=K))
    else:
        assert np.array_equal(Sp.n, np.bincount(lab_p, minlength=K)) and Sp.n.sum() == N
        lab_q, Sq = engine.gibbs_labels(c, b, W, seed=99, sweep=7)
        assert np.array_equal(lab_q, lab_p) and np.array_equal(Sq.sxx, Sp.sxx)


@pytest.mark.parametrize("D,K", [(2, 4), (2, 25), (3, 8), (4, 16), (1, 6)])
def test_small_shape_kernel_structures(engine, D, K):
    """Structure hints on the small-shape kernel: diagonal blocks return zero off-diagonal second moments, tied blocks
    ('linear') the pooled second moment and the bound with the shared quadratic term added back by the engine."""
    from oracle import mimo_oracle as O
    from scipy.special import logsumexp
    N = 20011
    rng = np.random.default_rng(31 * D + K)
    Z, c, b, W = _random_problem(rng, N, D, K)
    Wd = np.stack([np.diag(np.diag(w)) for w in W])
    engine.upload(Z)
    engine.set_structure('diag')
    try:
        L = O.canonical_eval(Z, c, b, Wd)
        lse = logsumexp(L, axis=0)
        n, sx, sxx = O.packed_stats(Z, np.exp(L - lse))
        S, sc = engine.estep(c, b, Wd)
        diag = np.stack([np.diag(np.diag(m)) for m in sxx])
        assert rel_err(S.n, n) < 1e-11 and rel_err(S.sx, sx) < 1e-11 and rel_err(S.sxx, diag) < 1e-11
        assert abs(sc[0] - lse.sum()) < 1e-12 * abs(lse.sum())
        if D > 1:
            with pytest.raises(ValueError):
                engine.estep(c, b, W)
        engine.set_structure('linear')
        Wt = np.stack(K * [W[0]])
        L = O.canonical_eval(Z, c, b, Wt)
        lse = logsumexp(L, axis=0)
        n, sx, sxx = O.packed_stats(Z, np.exp(L - lse))
        S, sc = engine.estep(c, b, Wt)
        assert S.sxx is None and rel_err(S.sxx_total, Z.T @ Z) < 1e-11
        assert rel_err(S.n, n) < 1e-11 and rel_err(S.sx, sx) < 1e-11
        assert abs(sc[0] - lse.sum()) < 1e-11 * abs(lse.sum())
    finally:
        engine.set_structure('full')


@pytest.mark.parametrize("D,K", [(8, 256), (8, 65), (8, 96), (8, 97), (5, 130), (9, 160), (9, 256), (1, 200), (3, 224), (3, 144), (3, 256), (3, 170), (4, 200),
                                 (7, 128), (2, 255), (6, 100), (8, 17), (8, 32), (5, 40), (9, 64), (3, 64), (7, 33), (6, 48),
                                 (8, 1), (8, 5), (7, 16), (5, 2), (9, 10), (6, 13),
                                 (16, 64), (12, 100), (10, 17), (16, 5), (13, 40), (12, 128), (11, 70), (15, 33), (14, 64), (10, 128)])
@pytest.mark.parametrize("N", [1, 15, 4099, 8 * 256 * 16 * 3 + 7])
def test_large_k_label_pass_and_label_statistics(engine, D, K, N):
    """The row-owner label kernel (Theta in LDS, draw in registers; every instantiation KB = 6..16) and the
    label-indexed statistics kernel (bitmap + popcount ranks, ascending rows per component) of mimo_rowwave.hip against
    the oracle: labels bit-exact for host uniforms and the Philox stream, counts exact, statistics to 1e-11, identical bits
    on a second launch; also with every row on ONE component and on three components (long per-component lists)."""
    from oracle import mimo_oracle as O
    rng = np.random.default_rng(900 + 10 * D + K)
    Z, c, b, W = _random_problem(rng, N, D, K)
    engine.upload(Z)
    mid = K <= 48 and (D >= 14 if K >= 33 else D >= 20 if K >= 17 else D >= 17)          # the mid kernel's label mode (round 4)
    assert engine.plan(K, gibbs=True)["kind"] == ("narrow" if (D <= 4 and 33 <= K <= 128) or (D <= 3 and 128 < K <= 256) or (D == 5 and K <= 64) or (5 <= D <= 16 and K <= (24 if D <= 8 else 16)) else "mid" if mid else "rowwave")
    L = O.canonical_eval(Z, c, b, W)
    u = rng.random(N)
    lab, S = engine.gibbs_labels(c, b, W, u=u)
    ref = O.sample_discrete_from_log(L, u)
    assert np.array_equal(lab, ref)
    n, sx, sxx = O.packed_stats(Z, O.one_hot(ref, K))
    assert np.array_equal(S.n, n) and rel_err(S.sx, sx) < 1e-11 and rel_err(S.sxx, sxx) < 1e-11
    lab_p, Sp = engine.gibbs_labels(c, b, W, seed=5, sweep=3)
    ref_p = O.sample_discrete_from_log(L, O.philox_uniforms(5, np.arange(N), 3))
    assert np.array_equal(lab_p, ref_p) and np.array_equal(Sp.n, np.bincount(ref_p, minlength=K))
    lab_q, Sq = engine.gibbs_labels(c, b, W, seed=5, sweep=3)
    assert np.array_equal(lab_q, lab_p) and np.array_equal(Sq.sxx, Sp.sxx) and np.array_equal(Sq.sx, Sp.sx)
    assert np.array_equal(engine.get_labels(), ref_p)
    for lab_c in (np.full(N, K - 1), rng.choice([0, K // 2, K - 1], size=N)):
        Sc = engine.label_stats(lab_c, K)
        nc, sxc, sxxc = O.packed_stats(Z, O.one_hot(lab_c, K))
        assert np.array_equal(Sc.n, nc) and rel_err(Sc.sx, sxc) < 1e-11 and rel_err(Sc.sxx, sxxc) < 1e-11
    # a component switched off by its weight never receives a label
    if K > 1:
        c2 = c.copy(); c2[K // 3] = -np.inf
        lab_o, So = engine.gibbs_labels(c2, b, W, seed=1, sweep=1)
        assert not np.any(lab_o == K // 3) and So.n[K // 3] == 0 and So.n.sum() == N


@pytest.mark.parametrize("D,K", [(32, 128), (20, 72), (24, 40), (17, 17), (32, 256), (28, 24), (16, 128), (13, 100), (10, 200),
                                 (16, 256), (12, 129), (31, 64), (23, 33)])
@pytest.mark.parametrize("N", [1, 4099, 20011])
def test_sliced_label_statistics(engine, D, K, N):
    """Label-indexed statistics where one launch cannot hold a component's accumulators (Dz = 17 .. 32, and K > 64 at
    Dz = 10 .. 16: label_stats_xwide_kernel, feature slices over several launches): after the label draw of the sweep
    and for caller-supplied labels, against the oracle — labels and counts exact, statistics to 1e-11, identical bits on
    a second launch, every row on one component / on three components."""
    from oracle import mimo_oracle as O
    rng = np.random.default_rng(1300 + 10 * D + K)
    Z, c, b, W = _random_problem(rng, N, D, K)
    engine.upload(Z)
    assert engine.plan(K, gibbs=True)["kind"] in ("two-stage", "fused", "rowwave", "mid")     # (rowwave: Theta streamed through LDS; mid: K <= 48)
    L = O.canonical_eval(Z, c, b, W)
    lab_p, Sp = engine.gibbs_labels(c, b, W, seed=9, sweep=2)
    ref_p = O.sample_discrete_from_log(L, O.philox_uniforms(9, np.arange(N), 2))
    assert np.array_equal(lab_p, ref_p)
    n, sx, sxx = O.packed_stats(Z, O.one_hot(ref_p, K))
    assert np.array_equal(Sp.n, n) and rel_err(Sp.sx, sx) < 1e-11 and rel_err(Sp.sxx, sxx) < 1e-11
    lab_q, Sq = engine.gibbs_labels(c, b, W, seed=9, sweep=2)
    assert np.array_equal(lab_q, lab_p) and np.array_equal(Sq.sxx, Sp.sxx) and np.array_equal(Sq.sx, Sp.sx)
    for lab_c in (np.full(N, K - 1), rng.choice([0, K // 2, K - 1], size=N), rng.integers(K, size=N)):
        Sc = engine.label_stats(lab_c, K)
        nc, sxc, sxxc = O.packed_stats(Z, O.one_hot(lab_c, K))
        assert np.array_equal(Sc.n, nc) and rel_err(Sc.sx, sxc) < 1e-11 and rel_err(Sc.sxx, sxxc) < 1e-11


def test_sample_discrete_from_log_function(engine):
    """mimo_amd.utils.stats.sample_discrete_from_log (mimo/utils/stats.py:8-21 of the reference) on a caller-supplied
    table: same draw as the reference's formula for the same numpy.random state, any axis, any leading shape, with and
    without the log-normalisers; the Philox variant; the resident-table variant of the engine."""
    import numpy.random as npr
    from scipy.special import logsumexp
    from oracle import mimo_oracle as O
    from mimo_amd.utils.stats import sample_discrete_from_log
    rng = np.random.default_rng(17)
    for shape, axis in (((7, 5000), 0), ((5000, 7), 1), ((3, 40, 11), 2), ((1, 9), 0), ((256, 300), 0), ((4, 0), 0)):
        p = rng.standard_normal(shape) * 3.
        npr.seed(5)
        got, ln = sample_discrete_from_log(p, return_lognorms=True, axis=axis, engine=engine)
        npr.seed(5)
        size = list(shape); size[axis] = 1
        u = npr.random(size=size)
        moved = np.moveaxis(p, axis, 0).reshape(shape[axis], -1)
        ref = O.sample_discrete_from_log(moved, u.reshape(-1)) if moved.shape[1] else np.zeros(0, np.int32)
        rest = tuple(s for i, s in enumerate(shape) if i != axis)
        assert got.dtype == np.int32 and got.shape == rest and np.array_equal(got.reshape(-1), ref)
        assert rel_err(ln, logsumexp(p, axis=axis)) < 1e-13
    p = rng.standard_normal((12, 3000))
    got = sample_discrete_from_log(p, seed=77, sweep=2, engine=engine)
    assert np.array_equal(got, O.sample_discrete_from_log(p, O.philox_uniforms(77, np.arange(3000), 2)))
    # a table left on the device by the E-step
    Z, c, b, W = _random_problem(rng, 4000, 3, 70)
    engine.upload(Z)
    engine.estep(c, b, W, stats=False, keep_logp=True)
    u = rng.random(4000)
    assert np.array_equal(engine.sample_from_log(u=u), O.sample_discrete_from_log(O.canonical_eval(Z, c, b, W), u))


@pytest.mark.parametrize("D,K,N", [(2, 4, 20000), (16, 64, 30011), (8, 200, 10007), (32, 16, 5000)])
def test_device_side_random_start(engine, D, K, N):
    """randomize=True without host uniforms: r[k,n] = v_kn / sum_j v_jn from the Philox stream (key seed, counter (row, k));
    statistics and the resident table against the oracle's restatement of the same stream; a driver run started this way
    equals the same run on the oracle-backed engine double."""
    from oracle import mimo_oracle as O
    from oracle_engine import OracleEngine
    rng = np.random.default_rng(D + K)
    Z, c, b, W = _random_problem(rng, N, D, K)
    engine.upload(Z)
    S = engine.random_resp_stats(K, seed=4242)
    V = np.stack([O.philox_uniforms(4242, np.arange(N), k) for k in range(K)]) + 1.1102230246251565e-16
    R = V / V.sum(axis=0)
    n, sx, sxx = O.packed_stats(Z, R)
    assert rel_err(S.n, n) < 1e-12 and rel_err(S.sx, sx) < 1e-11 and rel_err(S.sxx, sxx) < 1e-11
    assert rel_err(engine.get_resp(K), R) < 1e-14
    if D == 2:
        g = load_golden("gmm_c1_d2_k4_dir")
        traces = []
        for eng in (engine, OracleEngine()):
            kind, model = mc.build_gmm(g, eng)
            np.random.seed(3)
            traces.append(model.meanfield_coordinate_descent(g["X"], randomize=True, maxiter=8, tol=0., progress_bar=False,
                                                            init_rng='philox', seed=11))
        assert rel_err(traces[0], traces[1]) < 1e-10 and np.all(np.diff(traces[0]) > -1e-8 * abs(traces[0][-1]))


def test_lazy_log_prob_table(engine):
    """resample_labels returns the reference's (log_prob, labels) pair, the table as a LazyTable: nothing (K, N)-sized is
    computed or copied until it is used; then it is the table the labels were drawn from."""
    import numpy.random as npr
    from mimo_amd.mixtures.gmm import LazyTable
    g = load_golden("gibbs_c1_trace")
    kind, model = mc.build_gmm(g, engine)
    X = g["X"]
    npr.seed(9)
    lp, labels = model.resample_labels(X)
    assert isinstance(lp, LazyTable) and not lp.evaluated and lp.shape == (model.size, len(X))
    npr.seed(9)
    lp_eager, labels_eager = model.resample_labels(X, lazy=False)
    assert np.array_equal(labels, labels_eager)
    assert np.array_equal(np.asarray(lp), lp_eager) and lp.evaluated and np.array_equal(lp[1], lp_eager[1])
    assert rel_err(lp_eager, model.likelihood.log_complete_likelihood(X)) < 1e-14


@pytest.mark.parametrize("ordered", [False, True])
def test_native_rccl_communicator_one_rank(engine, ordered, monkeypatch):
    """mimo_comm_init: the library's own RCCL communicator (no torch.distributed).  One rank is all a one-GPU box can
    run: the collective is issued on the context's stream behind the kernels, so every entry point must return the very
    numbers it returns without it — synchronous, asynchronous, device-out, Gibbs and the small-shape kernel.  `ordered`:
    the rank-ordered sum (ncclAllGather + the in-order add kernel) forced on for the one rank; otherwise ncclAllReduce."""
    from mimo_amd.engine import HipEngine
    if ordered:
        monkeypatch.setenv("MIMO_COMM_RANK_ORDER_FORCE", "1")
    rng = np.random.default_rng(5)
    eng = HipEngine(0)
    try:
        for (N, D, K) in ((30011, 16, 64), (20000, 2, 4), (9000, 8, 200)):
            Z, c, b, W = _random_problem(rng, N, D, K)
            eng.upload(Z)
            S0, sc0 = eng.estep(c, b, W)
            lab0, G0 = eng.gibbs_labels(c, b, W, seed=3, sweep=1)
            eng.comm_init(HipEngine.comm_unique_id(), 0, 1)
            S1, sc1 = eng.estep(c, b, W)
            assert np.array_equal(S1.sxx, S0.sxx) and np.array_equal(S1.n, S0.n) and np.array_equal(sc1[:1], sc0[:1])
            eng.estep_async(c, b, W)
            S2, sc2 = eng.estep_wait()
            assert np.array_equal(S2.sxx, S0.sxx) and sc2[0] == sc0[0]
            lab1, G1 = eng.gibbs_labels(c, b, W, seed=3, sweep=1)
            assert np.array_equal(lab1, lab0) and np.array_equal(G1.sxx, G0.sxx)
            _, sc3 = eng.estep(c, b, W, stats=False)
            assert sc3[0] == sc0[0]
            eng.comm_destroy()
    finally:
        eng.close()


@pytest.mark.parametrize("name", ["nan_rows_gmm_d3_k5", "nan_rows_gmm_d16_k70"])
def test_rows_with_nan(engine, name):
    """NaN rows on the HIP path (small-shape kernel at D = 3, K = 5; tile kernels + row-owner label kernels at
    D = 16 / K = 70) against the reference's outputs; a borrowed device buffer is not modified."""
    mc.check_nan_rows(name, engine)
    import torch
    g = load_golden(name)
    t = torch.tensor(g["X"], device="cuda:0")
    engine.upload(t)
    assert engine.n_bad == len(g["bad"]) and torch.isnan(t).any()
    engine.upload(np.nan_to_num(g["X"]))
    assert engine.n_bad == 0


def test_rows_with_nan_in_a_linear_gaussian_mixture(engine):
    """lingauss.py:150-151 / ilr.py:71-75 of the reference: element-wise nan_to_num in the experts' density; reference fixture."""
    import model_checks as mc
    mc.check_nan_rows_ilr("nan_rows_ilr_dx2_dy1_k6", engine)


def test_row_weights_stay_resident(engine):
    """mimo_estep_weighted with MIMO_F_WEIGHTS_RESIDENT: the engine re-sends a weight vector only when its content changed;
    a label pass with host uniforms (which takes the same device buffer) or a new data set invalidates the resident copy."""
    from oracle import mimo_oracle as O
    from scipy.special import logsumexp
    rng = np.random.default_rng(8)
    N, D, K = 30011, 7, 20
    Z, c, b, W = _random_problem(rng, N, D, K)
    engine.upload(Z)
    L = O.canonical_eval(Z, c, b, W)
    R = np.exp(L - logsumexp(L, axis=0))
    w = rng.random(N) + 0.1

    def check(w):
        S, _ = engine.estep(c, b, W, row_weights=w)
        n, sx, sxx = O.packed_stats(Z, R * w)
        assert rel_err(S.n, n) < 1e-11 and rel_err(S.sxx, sxx) < 1e-11
    check(w)
    assert engine._w_key is not None
    check(w)                                            # resident copy
    w *= 0.5                                            # edited in place: new fingerprint, re-sent
    check(w)
    engine.gibbs_labels(c, b, W, u=rng.random(N), stats=False)
    assert engine._w_key is None
    check(w)
    engine.upload(Z[:20000]); engine.upload(Z)
    check(w)


@pytest.mark.parametrize("D,K", [(8, 64), (8, 49), (8, 32), (8, 17), (8, 16), (8, 3), (5, 64), (5, 7), (7, 50), (9, 32), (9, 20),
                                 (6, 1), (3, 24), (4, 20), (2, 33), (1, 33)])
@pytest.mark.parametrize("N", [1, 17, 5003, 8 * 256 * 16 * 2 + 11])
def test_row_owner_softmax_pass(engine, D, K, N):
    """vi_rowwave_kernel (K <= 64, Dz <= 9: both matrix products on row-owner waves, r through a wave-private LDS block):
    statistics and sum_n lse_n against the oracle, identical bits on a second launch, asynchronous form; the generic
    requests (tables, entropy split, row weights) of the same shape still go through the tile kernels and agree."""
    from oracle import mimo_oracle as O
    from scipy.special import logsumexp
    rng = np.random.default_rng(4000 + 10 * D + K)
    Z, c, b, W = _random_problem(rng, N, D, K)
    engine.upload(Z)
    narrow = (D, K) in ((2, 33), (1, 33)) or (5 <= D <= 16 and K <= (24 if D <= 8 else 16)) or (D == 5 and K <= 64) or (D <= 4 and K >= (17 if D <= 2 else 9 if D == 3 else 12))   # (few components at Dz >= 5: the table-driven narrow kernels)
    engine.tune("mid_min_d", 64)          # (K = 33 .. 48 at Dz = 9 goes to the mid kernels by default since round 4)
    assert engine.plan(K)["kind"] == ("narrow" if narrow else "rowwave-vi")
    L = O.canonical_eval(Z, c, b, W)
    lse = logsumexp(L, axis=0)
    R = np.exp(L - lse)
    n, sx, sxx = O.packed_stats(Z, R)
    S, sc = engine.estep(c, b, W)
    assert rel_err(S.n, n) < 1e-11 and rel_err(S.sx, sx) < 1e-11 and rel_err(S.sxx, sxx) < 1e-11
    assert abs(sc[0] - lse.sum()) < 1e-12 * max(1., abs(lse.sum()))
    S1, sc1 = engine.estep(c, b, W)
    assert np.array_equal(S1.sxx, S.sxx) and np.array_equal(S1.sx, S.sx) and np.array_equal(S1.n, S.n) and sc1[0] == sc[0]
    engine.estep_async(c, b, W)
    S2, sc2 = engine.estep_wait()
    assert np.array_equal(S2.sxx, S.sxx) and sc2[0] == sc[0]
    if N <= 6000:
        Sg, scg = engine.estep(c, b, W, keep_resp=True, entropy_split=True)
        assert rel_err(Sg.sxx, sxx) < 1e-11 and rel_err(engine.get_resp(), R) < 1e-11 and abs(scg[0] - sc[0]) < 1e-11 * max(1., abs(sc[0]))
        c2 = c.copy(); c2[K // 2] = -np.inf                          # a switched-off component
        if K > 1:
            L2 = L.copy(); L2[K // 2] = -np.inf
            lse2 = logsumexp(L2, axis=0)
            n2, _, sxx2 = O.packed_stats(Z, np.exp(L2 - lse2))
            So, sco = engine.estep(c2, b, W)
            assert So.n[K // 2] < 1e-290 and rel_err(So.sxx, sxx2) < 1e-11 and abs(sco[0] - lse2.sum()) < 1e-12 * max(1., abs(lse2.sum()))


def test_empty_data_on_the_row_owner_kernels(engine):
    """N = 0 through the row-owner routes: zero statistics, no labels, no launch that reads a row."""
    rng = np.random.default_rng(1)
    for D, K in ((8, 32), (8, 200), (12, 40)):
        Z, c, b, W = _random_problem(rng, 0, D, K)
        engine.upload(Z)
        S, sc = engine.estep(c, b, W)
        assert not S.n.any() and not S.sxx.any() and sc[0] == 0.0
        lab, G = engine.gibbs_labels(c, b, W, seed=1, sweep=1)
        assert lab.shape == (0,) and not G.n.any() and not G.sxx.any()
        assert not engine.label_stats(np.zeros(0, dtype=np.int32), K).n.any()


NARROW_SHAPES = [(2, 50), (2, 64), (2, 33), (2, 100), (2, 128), (1, 50), (1, 97), (1, 128), (3, 50), (3, 64), (3, 96), (3, 127),
                 (4, 48), (4, 50), (4, 64), (4, 100), (4, 128), (2, 37), (2, 41), (2, 53), (2, 57), (2, 69), (2, 77), (2, 85),
                 (2, 93), (2, 101), (2, 109), (2, 117), (2, 125),
                 # 129 .. 256 components over at most two contraction steps (mimo_narrow_big.hip: one slot count per 16-component band)
                 (2, 129), (2, 144), (2, 150), (2, 176), (2, 192), (2, 200), (2, 224), (2, 240), (2, 256), (1, 160), (1, 256)]


# ... and few components over many features (Dz = 5 .. 16, K <= 16: the table-driven loops; every contraction length once)
NARROW_WIDE_SHAPES = [(5, 16), (5, 24), (8, 23), (6, 17), (6, 8), (7, 12), (8, 4), (8, 16), (9, 7), (10, 8), (11, 16), (12, 4), (12, 13), (13, 9), (14, 12),
                      (15, 5), (15, 8), (16, 1), (16, 4), (16, 8), (9, 16), (6, 3), (16, 2), (14, 2),
                      # Dz = 17 .. 32, K <= 8 (softmax pass only; the accumulators in the second half of the register file)
                      (17, 4), (18, 7), (19, 3), (20, 8), (21, 2), (22, 5), (23, 8), (24, 4), (25, 6), (26, 8), (27, 4), (28, 1),
                      (29, 3), (30, 4), (31, 2), (32, 4), (24, 7)]


@pytest.mark.parametrize("D,K", NARROW_SHAPES + NARROW_WIDE_SHAPES)
@pytest.mark.parametrize("N", [1, 17, 5003, 4 * 256 * 16 * 5 + 11])
def test_narrow_kernels_vs_oracle(engine, D, K, N):
    """narrow_kernel (mimo_narrow.hip: Dz <= 4 with 32 < K <= 128 — the reference's ILR defaults, examples/ilr/evaluate_sine.py:35 —
    both products on v_mfma_f64_4x4x4_4b, every instantiated slot count V): the softmax pass (statistics, sum_n lse_n, identical
    bits on a second launch, asynchronous form, a switched-off component) and the label pass (labels bit-exact for host
    uniforms and the Philox stream, counts exact, statistics) against the oracle; the generic requests of the same shape
    still go through the tile kernels and agree."""
    from oracle import mimo_oracle as O
    from scipy.special import logsumexp
    rng = np.random.default_rng(7000 + 10 * D + K)
    Z, c, b, W = _random_problem(rng, N, D, K)
    engine.upload(Z)
    engine.tune("mid_min_d", 64)          # (the mid kernels are preferred for some of these shapes since round 4: every narrow instantiation stays tested)
    engine.tune("narrow_big_vi", 256)     # (... and the softmax pass of 193 .. 256 components over two steps stays on the tile kernels by default)
    assert engine.plan(K)["kind"] == "narrow" and (D > 16 or engine.plan(K, gibbs=True)["kind"] == "narrow")
    L = O.canonical_eval(Z, c, b, W)
    lse = logsumexp(L, axis=0)
    R = np.exp(L - lse)
    n, sx, sxx = O.packed_stats(Z, R)
    S, sc = engine.estep(c, b, W)
    assert rel_err(S.n, n) < 1e-11 and rel_err(S.sx, sx) < 1e-11 and rel_err(S.sxx, sxx) < 1e-11
    assert abs(sc[0] - lse.sum()) < 1e-12 * max(1., abs(lse.sum()))
    S1, sc1 = engine.estep(c, b, W)
    assert np.array_equal(S1.sxx, S.sxx) and np.array_equal(S1.sx, S.sx) and np.array_equal(S1.n, S.n) and sc1[0] == sc[0]
    engine.estep_async(c, b, W)
    S2, sc2 = engine.estep_wait()
    assert np.array_equal(S2.sxx, S.sxx) and sc2[0] == sc[0]
    # per-row weights of the statistics ride on the normaliser (mimo_estep_weighted on the same kernels): scalars unweighted
    w = rng.uniform(0., 2., size=N)
    wn, wsx, wsxx = O.packed_stats(Z, R * w[None, :])
    Sw, scw = engine.estep(c, b, W, row_weights=w)
    assert rel_err(Sw.n, wn) < 1e-11 and rel_err(Sw.sx, wsx) < 1e-11 and rel_err(Sw.sxx, wsxx) < 1e-11
    assert abs(scw[0] - sc[0]) <= 1e-13 * max(1., abs(sc[0]))
    # label pass
    u = rng.random(N)
    lab, G = engine.gibbs_labels(c, b, W, u=u)
    ref = O.sample_discrete_from_log(L, u)
    assert np.array_equal(lab, ref)
    gn, gsx, gsxx = O.packed_stats(Z, O.one_hot(ref, K))
    assert np.array_equal(G.n, gn) and rel_err(G.sx, gsx) < 1e-11 and rel_err(G.sxx, gsxx) < 1e-11
    lab_p, Gp = engine.gibbs_labels(c, b, W, seed=5, sweep=3)
    ref_p = O.sample_discrete_from_log(L, O.philox_uniforms(5, np.arange(N), 3))
    assert np.array_equal(lab_p, ref_p) and np.array_equal(Gp.n, np.bincount(ref_p, minlength=K))
    lab_q, Gq = engine.gibbs_labels(c, b, W, seed=5, sweep=3)
    assert np.array_equal(lab_q, lab_p) and np.array_equal(Gq.sxx, Gp.sxx)
    assert np.array_equal(engine.get_labels(), ref_p)
    if N <= 6000:
        Sg, scg = engine.estep(c, b, W, keep_resp=True, entropy_split=True)      # generic request: tile kernels
        assert rel_err(Sg.sxx, sxx) < 1e-11 and rel_err(engine.get_resp(), R) < 1e-11 and abs(scg[0] - sc[0]) < 1e-11 * max(1., abs(sc[0]))
        if K == 1:
            return
        c2 = c.copy(); c2[K // 2] = -np.inf                          # a switched-off component
        L2 = L.copy(); L2[K // 2] = -np.inf
        lse2 = logsumexp(L2, axis=0)
        n2, _, sxx2 = O.packed_stats(Z, np.exp(L2 - lse2))
        So, sco = engine.estep(c2, b, W)
        assert So.n[K // 2] < 1e-290 and rel_err(So.sxx, sxx2) < 1e-11 and abs(sco[0] - lse2.sum()) < 1e-12 * max(1., abs(lse2.sum()))
        lab_o, Go = engine.gibbs_labels(c2, b, W, seed=1, sweep=1)
        assert not np.any(lab_o == K // 2) and Go.n[K // 2] == 0 and Go.n.sum() == N


@pytest.mark.parametrize("D,K", [(16, 128), (12, 200), (20, 72), (16, 256), (32, 40), (24, 128),
                                 (28, 200), (32, 210), (25, 224)])     # 14 row blocks do not fit next to the z rows: run with 16
def test_streamed_label_kernel_many_steps(engine, D, K):
    """gibbs_stream_kernel (the row-owner label kernel with Theta streamed through a double buffer in LDS: shapes whose operand
    image does not fit) over several workgroup steps per workgroup — the cyclic walk of the chunks across step boundaries, both
    buffer parities: labels bit-exact for host uniforms and the Philox stream, statistics of the sweep, identical bits on a
    second launch.  (tests of one step per workgroup: test_sliced_label_statistics.)"""
    from oracle import mimo_oracle as O
    N = 128 * 256 * 2 + 77 if K * D * D < 150000 else 128 * 256 + 515        # (the oracle's share of the three largest shapes)
    rng = np.random.default_rng(2100 + 10 * D + K)
    Z, c, b, W = _random_problem(rng, N, D, K)
    engine.upload(Z)
    engine.tune("mid_labels_min_d", 64)      # ((32, 40) runs on the mid kernel's label mode by default since round 4: the streamed kernel stays tested)
    assert engine.plan(K, gibbs=True)["kind"] == "rowwave"
    L = O.canonical_eval(Z, c, b, W)
    u = rng.random(N)
    lab, S = engine.gibbs_labels(c, b, W, u=u)
    ref = O.sample_discrete_from_log(L, u)
    assert np.array_equal(lab, ref)
    n, sx, sxx = O.packed_stats(Z, O.one_hot(ref, K))
    assert np.array_equal(S.n, n) and rel_err(S.sx, sx) < 1e-11 and rel_err(S.sxx, sxx) < 1e-11
    lab_p, Sp = engine.gibbs_labels(c, b, W, seed=3, sweep=7)
    ref_p = O.sample_discrete_from_log(L, O.philox_uniforms(3, np.arange(N), 7))
    assert np.array_equal(lab_p, ref_p)
    lab_q, Sq = engine.gibbs_labels(c, b, W, seed=3, sweep=7)
    assert np.array_equal(lab_q, lab_p) and np.array_equal(Sq.sxx, Sp.sxx)


@pytest.mark.parametrize("seed", [3])
def test_random_shapes_structures_and_sizes_through_both_passes(engine, seed):
    """tools/fuzz_parity.py in small: 50 random (Dz, K, structure, N) through the softmax pass (plain, weighted,
    asynchronous, second launch) and the label pass (host uniforms, Philox, second launch) against the oracle — whatever
    kernel family the router picks for the shape (a guard for the borders between the families)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity
    assert fuzz_parity.run(50, seed, eng=engine, max_rows=33000) == 0


@pytest.mark.parametrize("D,K", [(2, 50), (3, 12), (8, 4), (16, 8), (20, 4), (5, 40), (20, 40), (32, 100), (24, 12)])
def test_rows_with_nan_on_the_narrow_kernels(engine, D, K):
    """Rows that hold a NaN (section 4c of DESIGN.md: gaussian.py:493-494,512-520 of the reference) on the narrow kernels: the
    softmax pass takes the row mask as its per-row weights (statistics without the rows, sum_n lse_n with them at z = 0), the
    label pass draws every label and takes the statistics of the valid rows."""
    from oracle import mimo_oracle as O
    from scipy.special import logsumexp
    rng = np.random.default_rng(8800 + 10 * D + K)
    N = 20011
    Z, c, b, W = _random_problem(rng, N, D, K)
    bad = rng.choice(N, size=37, replace=False)
    Zn = Z.copy()
    Zn[bad, rng.integers(0, D, size=bad.size)] = np.nan
    mask = np.ones(N); mask[bad] = 0.
    Zc = Z.copy(); Zc[bad] = 0.
    engine.upload(Zn)
    # ((20, 40), (32, 100), (24, 12): streamed label kernel + one-pass label statistics over the masked labels, two-stage softmax pass)
    assert engine.n_bad == bad.size and (engine.plan(K)["kind"] == "narrow" or (D > 16 and K > 8))
    L = O.canonical_eval(Zc, c, b, W)
    lse = logsumexp(L, axis=0)
    R = np.exp(L - lse)
    n, sx, sxx = O.packed_stats(Zc, R * mask[None, :])
    S, sc = engine.estep(c, b, W)
    assert rel_err(S.n, n) < 1e-11 and rel_err(S.sx, sx) < 1e-11 and rel_err(S.sxx, sxx) < 1e-11
    assert abs(sc[0] - lse.sum()) < 1e-12 * max(1., abs(lse.sum()))
    lab, G = engine.gibbs_labels(c, b, W, seed=4, sweep=2)
    ref = O.sample_discrete_from_log(L, O.philox_uniforms(4, np.arange(N), 2))
    assert np.array_equal(lab, ref)
    gn, gsx, gsxx = O.packed_stats(Zc, O.one_hot(ref, K) * mask[None, :])
    assert np.array_equal(G.n, gn) and rel_err(G.sx, gsx) < 1e-11 and rel_err(G.sxx, gsxx) < 1e-11
    engine.upload(Z)
    assert engine.n_bad == 0


@pytest.mark.parametrize("D,K", [(8, 256), (2, 64), (9, 40), (4, 128), (6, 17)])
def test_label_kernels_that_count_their_labels(engine, D, K):
    """N >= 2^17 rows, K >= 17, Dz <= 9: the statistics of the drawn labels come from label_stats_slots_kernel, whose slot table
    is built from a histogram the label kernel counted itself (KernelArgs::fuse_hist: gibbs_rowwave_kernel, narrow_kernel) —
    labels bit-exact, counts exact, statistics against the oracle, a second sweep bit-identical; the statistics of a caller's
    label vector (histogram by label_hist_kernel) agree bit for bit with those of the sweep that drew the same labels (where the
    sweep uses the same statistics kernel)."""
    from oracle import mimo_oracle as O
    N = 140003
    rng = np.random.default_rng(9900 + 10 * D + K)
    Z, c, b, W = _random_problem(rng, N, D, K)
    c = c + 3. * rng.standard_normal(K)                    # uneven component weights: uneven slot table
    engine.upload(Z)
    L = O.canonical_eval(Z, c, b, W)
    lab, G = engine.gibbs_labels(c, b, W, seed=6, sweep=1)
    ref = O.sample_discrete_from_log(L, O.philox_uniforms(6, np.arange(N), 1))
    assert np.array_equal(lab, ref)
    gn, gsx, gsxx = O.packed_stats(Z, O.one_hot(ref, K))
    assert np.array_equal(G.n, gn) and rel_err(G.sx, gsx) < 1e-11 and rel_err(G.sxx, gsxx) < 1e-11
    lab2, G2 = engine.gibbs_labels(c, b, W, seed=6, sweep=1)
    assert np.array_equal(lab2, lab) and np.array_equal(G2.sxx, G.sxx) and np.array_equal(G2.sx, G.sx)
    S = engine.label_stats(lab, K)
    assert np.array_equal(S.n, G.n) and rel_err(S.sxx, G.sxx) < 1e-12
    if (D, K) != (6, 17):              # (6, 17): the sweep takes its statistics inside the narrow label kernel — another summation order
        assert np.array_equal(S.sxx, G.sxx)


@pytest.mark.parametrize("D,K", [(2, 50), (1, 100), (2, 160), (4, 64), (8, 4), (16, 4), (32, 4), (20, 16), (17, 9), (24, 32), (8, 64), (16, 64), (3, 8)])
@pytest.mark.parametrize("N", [1, 5003, 70001])
def test_bound_only_pass(engine, D, K, N):
    """MIMO_F_NO_STATS without tables — the full-data bound of every SVI outer iteration (gmm.py:319-326, ilr.py:270-277 of the
    reference): sum_n lse_n against the oracle, equal to the scalar of the full pass of the same shape, synchronous and asynchronous
    form, with rows that hold a NaN; on the narrow kernels and on the mid kernels up to K = 16 it runs as the plain pass of the
    shape's own family (mimo_estep: bound_promote), elsewhere as the generic request on the tile kernels."""
    from oracle import mimo_oracle as O
    from scipy.special import logsumexp
    rng = np.random.default_rng(31000 + 10 * D + K)
    Z, c, b, W = _random_problem(rng, N, D, K)
    engine.upload(Z)
    want = float(logsumexp(O.canonical_eval(Z, c, b, W), axis=0).sum())
    S, sc_full = engine.estep(c, b, W)
    none, sc = engine.estep(c, b, W, stats=False)
    assert none is None and abs(sc[0] - want) < 1e-12 * max(1., abs(want)) and abs(sc[0] - sc_full[0]) <= 1e-13 * max(1., abs(want))
    engine.estep_async(c, b, W, stats=False)
    none, sc2 = engine.estep_wait()
    assert none is None and sc2[0] == sc[0]
    S2, sc3 = engine.estep(c, b, W)                              # the next full pass is not disturbed by the statistics left behind
    assert np.array_equal(S2.sxx, S.sxx) and sc3[0] == sc_full[0]
    if N > 100:
        Zn = Z.copy()
        bad = rng.choice(N, size=7, replace=False)
        Zn[bad, rng.integers(0, D, size=7)] = np.nan
        engine.upload(Zn)
        Ln = O.canonical_eval(np.nan_to_num(Zn) * (~np.isnan(Zn).any(axis=1))[:, None], c, b, W)
        _, scn = engine.estep(c, b, W, stats=False)
        _, scn_full = engine.estep(c, b, W)
        assert abs(scn[0] - scn_full[0]) <= 1e-12 * max(1., abs(scn_full[0]))
