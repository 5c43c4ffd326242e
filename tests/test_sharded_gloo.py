"""CPU, world_size 2 over gloo: the N>1 path (row shards + one all-reduce of the statistic block per
sweep) gives every rank the same global statistics / ELBO / posterior as the single-process run, and
Philox labels do not depend on the number of shards."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT, load_golden


def _worker(rank, world, port, name, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle_engine import OracleEngine
        from mimo_amd.sharded import ShardedEngine, shard_rows
        import model_checks as mc
        g = load_golden(name)
        X = g["X"]
        lo, hi = shard_rows(len(X), rank, world)
        eng = ShardedEngine(OracleEngine(), row_offset=lo)
        kind, model = mc.build_gmm(g, eng)
        mc.load_gmm_state(model, g, kind)
        Xl = np.ascontiguousarray(X[lo:hi])
        model._bind(Xl)
        c, b, W = model.likelihood.canonical()            # point estimates of the fixture
        labels, Sl = eng.gibbs_labels(c, b, W, seed=1337, sweep=3)
        vlb = model.meanfield_coordinate_descent(Xl, randomize=False, maxiter=len(g["vi_vlb"]), tol=0.,
                                                 progress_bar=False)
        S, sc = eng.estep(*model.canonical_expected())
        q.put((rank, np.array(vlb), model.components.posterior.mus.copy(), S.sxx.copy(), lo, labels, Sl.n.copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["gmm_c3_d8_k32_stick", "gmm_tail_d5_k7_stick"])
def test_two_rank_vi_matches_single_process(name):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = load_golden(name)
    # both ranks agree with each other and with the reference's single-process ELBO trace
    assert np.array_equal(res[0][1], res[1][1])
    assert np.max(np.abs(res[0][1] - g["vi_vlb"]) / np.abs(g["vi_vlb"])) < 1e-8
    assert np.allclose(res[0][2], g["vi_post_mus"], rtol=1e-6, atol=1e-9)
    assert np.array_equal(res[0][3], res[1][3])
    # labels: global-row Philox counters => concatenated shards equal the unsharded draw
    from oracle_engine import OracleEngine
    import model_checks as mc
    kind, model = mc.build_gmm(g, OracleEngine())
    mc.load_gmm_state(model, g, kind)
    eng = model._bind(g["X"])
    ref_labels, _ = eng.gibbs_labels(*model.likelihood.canonical(), seed=1337, sweep=3, stats=False)
    assert np.array_equal(np.concatenate([res[0][5], res[1][5]]), ref_labels)
    assert np.array_equal(res[0][6], np.bincount(ref_labels, minlength=int(g["K"])))


def _worker_hier(rank, world, port, name, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy.random as npr
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle_engine import OracleEngine
        from mimo_amd.sharded import ShardedEngine, shard_rows
        from mimo_amd.distributions import (Dirichlet, CategoricalWithDirichlet, NormalWishart, TiedGaussiansWithScaledPrecision,
                                            TiedGaussiansWithHierarchicalNormalWisharts)
        from mimo_amd.mixtures import BayesianMixtureOfGaussiansWithHierarchicalPrior
        g = load_golden(name)
        X, w = g["X"], g["w"]
        K, D, seed, iters, sub = (int(g[k]) for k in ("K", "D", "seed", "iters", "sub"))
        lo, hi = shard_rows(len(X), rank, world)
        eng = ShardedEngine(OracleEngine(), row_offset=lo)
        npr.seed(seed + 1)            # identical host state on every rank
        gating = CategoricalWithDirichlet(dim=K, prior=Dirichlet(dim=K, alphas=np.ones((K,))))
        hyper = NormalWishart(dim=D, mu=np.zeros((D,)), kappa=1e-2, psi=np.eye(D), nu=D + 1. + 1e-8)
        prior = TiedGaussiansWithScaledPrecision(size=K, dim=D, kappas=1e-2 * np.ones((K,)))
        comps = TiedGaussiansWithHierarchicalNormalWisharts(size=K, dim=D, hyper_prior=hyper, prior=prior, engine=eng)
        m = BayesianMixtureOfGaussiansWithHierarchicalPrior(size=K, dim=D, gating=gating, components=comps, engine=eng)
        # start both the sharded and the reference run from the fixture's post-Gibbs state
        c = m.components
        c.posterior.mus, c.posterior.kappas, c.posterior.lmbdas = g["gibbs_post_mus"], g["gibbs_post_kappas"], g["gibbs_post_lmbdas"]
        c.hyper_posterior.params = (g["gibbs_hyper_mu"], float(g["gibbs_hyper_kappa"]), g["gibbs_hyper_psi"], float(g["gibbs_hyper_nu"]))
        m.gating.posterior.alphas = g["gibbs_galphas"].copy()
        Xl, wl = np.ascontiguousarray(X[lo:hi]), np.ascontiguousarray(w[lo:hi])
        vlb = m.meanfield_coordinate_descent(Xl, randomize=False, weights=wl, maxiter=3, maxsubiter=sub, tol=0., progress_bar=False)
        S, _ = eng.estep(*m.canonical_expected())
        q.put((rank, np.array(vlb), c.posterior.mus.copy(), c.hyper_posterior.wishart.psi.copy(), eng._structure,
               S.sxx is None, S.sxx_total.copy()))
    finally:
        dist.destroy_process_group()


def test_two_rank_weighted_hierarchical_vi_matches_single_process():
    """Row weights and the structure hint pass through the sharded engine: a hierarchical GMM with per-row weights on
    two ranks reproduces the single-process run (bound trace, posterior means, pooled precision)."""
    name = "hier_gmm_d2_k4_m2"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_hier, args=(r, 2, port, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    # tied blocks travel on the 'linear' structure: no per-component second moments, the pooled one summed over the shards
    assert res[0][4] == 'linear' and res[0][5] and np.array_equal(res[0][6], res[1][6])
    assert np.allclose(res[0][6], load_golden(name)["X"].T @ load_golden(name)["X"], rtol=1e-12)
    # single-process reference with the same host code and the oracle engine
    import numpy.random as npr
    from oracle_engine import OracleEngine
    from mimo_amd.distributions import (Dirichlet, CategoricalWithDirichlet, NormalWishart, TiedGaussiansWithScaledPrecision,
                                        TiedGaussiansWithHierarchicalNormalWisharts)
    from mimo_amd.mixtures import BayesianMixtureOfGaussiansWithHierarchicalPrior
    g = load_golden(name)
    K, D, seed, sub = (int(g[k]) for k in ("K", "D", "seed", "sub"))
    eng = OracleEngine()
    npr.seed(seed + 1)
    gating = CategoricalWithDirichlet(dim=K, prior=Dirichlet(dim=K, alphas=np.ones((K,))))
    hyper = NormalWishart(dim=D, mu=np.zeros((D,)), kappa=1e-2, psi=np.eye(D), nu=D + 1. + 1e-8)
    prior = TiedGaussiansWithScaledPrecision(size=K, dim=D, kappas=1e-2 * np.ones((K,)))
    comps = TiedGaussiansWithHierarchicalNormalWisharts(size=K, dim=D, hyper_prior=hyper, prior=prior, engine=eng)
    m = BayesianMixtureOfGaussiansWithHierarchicalPrior(size=K, dim=D, gating=gating, components=comps, engine=eng)
    c = m.components
    c.posterior.mus, c.posterior.kappas, c.posterior.lmbdas = g["gibbs_post_mus"], g["gibbs_post_kappas"], g["gibbs_post_lmbdas"]
    c.hyper_posterior.params = (g["gibbs_hyper_mu"], float(g["gibbs_hyper_kappa"]), g["gibbs_hyper_psi"], float(g["gibbs_hyper_nu"]))
    m.gating.posterior.alphas = g["gibbs_galphas"].copy()
    vlb = m.meanfield_coordinate_descent(g["X"], randomize=False, weights=g["w"], maxiter=3, maxsubiter=sub, tol=0., progress_bar=False)
    assert np.max(np.abs(res[0][1] - np.array(vlb)) / np.abs(np.array(vlb))) < 1e-10
    assert np.allclose(res[0][2], c.posterior.mus, rtol=1e-9, atol=1e-12)
    assert np.allclose(res[0][3], c.hyper_posterior.wishart.psi, rtol=1e-9, atol=1e-14)


def _worker_svi(rank, world, port, name, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import random
    import numpy.random as npr
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle_engine import OracleEngine
        from mimo_amd.sharded import ShardedEngine, shard_rows
        import model_checks as mc
        g = load_golden(name)
        X = g["X"][:len(g["X"]) // 2 * 2]
        lo, hi = shard_rows(len(X), rank, world)
        eng = ShardedEngine(OracleEngine(), row_offset=lo)
        kind, model = mc.build_gmm(g, eng)
        mc.load_gmm_state(model, g, kind)
        npr.seed(11); random.seed(12)              # identical host streams on every rank
        vlb = model.meanfield_stochastic_descent(np.ascontiguousarray(X[lo:hi]), randomize=False, maxiter=3,
                                                 step_size=5e-2, batch_size=32, progress_bar=False)
        # row-local prediction passes through the sharded engine (second ADVICE item): moments of this rank's rows
        eng.upload(np.ascontiguousarray(X[lo:hi, :1]))
        K = model.size
        rng = np.random.default_rng(3)
        c, b, W = rng.standard_normal(K), rng.standard_normal((K, 1)), np.abs(rng.standard_normal((K, 1, 1))) + .5
        M, Q, Cc = rng.standard_normal((K, 1, 2)), np.stack(K * [np.eye(2)]), np.ones((K, 1, 1))
        mu, covar, _ = eng.predict(c, b, W, M, Q, Cc)
        q.put((rank, np.array(vlb), [p.copy() for p in model.components.posterior.params],
               model.gating.posterior.params, mu, type(model._batch_engine).__name__))
    finally:
        dist.destroy_process_group()


def test_two_rank_svi_and_predict():
    """meanfield_stochastic_descent on a ShardedEngine (it used to build ShardedEngine(<int>) and crash): the
    minibatch is the union of the ranks' local draws, its statistics are all-reduced, and scale = union / all rows —
    equal to a single process that takes the same union as its minibatch.  predict() is a row-local pass-through."""
    import random
    import numpy.random as npr
    name = "gmm_tail_d5_k7_stick"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_svi, args=(r, 2, port, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][1], res[1][1]) and res[0][5] == "ShardedEngine"
    # single-process emulation: same local index draw applied to both halves
    from oracle_engine import OracleEngine
    from mimo_amd.utils.data import batches
    from mimo_amd.mixtures.gmm import _component_stats
    import model_checks as mc
    g = load_golden(name)
    X = g["X"][:len(g["X"]) // 2 * 2]
    n_loc = len(X) // 2
    eng, beng = OracleEngine(), OracleEngine()
    kind, model = mc.build_gmm(g, eng)
    mc.load_gmm_state(model, g, kind)
    eng.upload(X)
    npr.seed(11); random.seed(12)
    scale, vlb = 2 * 32 / float(len(X)), []
    for _ in range(3):
        for batch in batches(32, n_loc):
            beng.upload(np.concatenate([X[:n_loc][batch], X[n_loc:][batch]]))
            Sb, _ = beng.estep(*model.canonical_expected())
            model.components.meanfield_sgd(None, None, scale, 5e-2, stats=_component_stats(Sb, model.components))
            model.gating.meanfield_sgd(None, Sb.n, scale, 5e-2)
        _, sc = eng.estep(*model.canonical_expected(), stats=False)
        vlb.append(model._vlb_prior_terms() + sc[0])
    assert np.max(np.abs(res[0][1] - np.array(vlb)) / np.abs(np.array(vlb))) < 1e-10
    for a, b in zip(res[0][2], model.components.posterior.params):
        assert np.allclose(a, b, rtol=1e-9, atol=1e-12)
    # prediction: concatenated shards equal the unsharded call
    K = model.size
    rng = np.random.default_rng(3)
    c, b, W = rng.standard_normal(K), rng.standard_normal((K, 1)), np.abs(rng.standard_normal((K, 1, 1))) + .5
    M, Q, Cc = rng.standard_normal((K, 1, 2)), np.stack(K * [np.eye(2)]), np.ones((K, 1, 1))
    one = OracleEngine()
    one.upload(np.ascontiguousarray(X[:, :1]))
    mu, _, _ = one.predict(c, b, W, M, Q, Cc)
    assert np.allclose(np.concatenate([res[0][4], res[1][4]]), mu, rtol=1e-12, atol=1e-14)


def _worker_order(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle_engine import OracleEngine
        from mimo_amd.sharded import ShardedEngine
        eng = ShardedEngine(OracleEngine())
        # blocks whose sum depends on the association: magnitudes 1e16 / 1 / -1e16
        blocks = [np.array([1e16, 1.0, 3.0]), np.array([1.0, 1e16, -1e16]), np.array([-1e16, -1e16, 1e16 + 2.0])]
        out = eng._allreduce_array(blocks[rank])
        # replication check: identical arrays pass, a rank-dependent array raises on every rank
        eng.assert_replicated(np.arange(5.0), np.ones((2, 2)))
        try:
            eng.assert_replicated(np.arange(5.0) + rank)
            raised = False
        except RuntimeError as exc:
            raised = "differ between the ranks" in str(exc)
        q.put((rank, out, raised))
    finally:
        dist.destroy_process_group()


def test_three_rank_sum_is_taken_in_rank_order_and_divergence_is_caught():
    """The sum over the ranks is ((block 0 + block 1) + block 2) on every rank, bit for bit (all-gather + ordered add: the
    association does not depend on the transport), and `assert_replicated` raises on every rank when the host-drawn
    parameters differ between the ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_order, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    blocks = [np.array([1e16, 1.0, 3.0]), np.array([1.0, 1e16, -1e16]), np.array([-1e16, -1e16, 1e16 + 2.0])]
    expect = (blocks[0] + blocks[1]) + blocks[2]
    assert not np.array_equal(expect, blocks[0] + (blocks[1] + blocks[2]))        # the test vectors do tell the orders apart
    for r in range(3):
        assert np.array_equal(res[r][1], expect) and res[r][2]


def _worker_nan(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle_engine import OracleEngine
        from mimo_amd.sharded import ShardedEngine, shard_rows
        rng = np.random.default_rng(5)
        N, D, K = 400, 3, 5
        X = rng.standard_normal((N, D)) * 2.
        X[[210, 333, 399], 1] = np.nan                  # rows with NaN on the SECOND shard only
        A = rng.standard_normal((K, D, D)); W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D)
        b, c = rng.standard_normal((K, D)), rng.standard_normal(K)
        lo, hi = shard_rows(N, rank, world)
        eng = ShardedEngine(OracleEngine(), row_offset=lo)
        eng.upload(np.ascontiguousarray(X[lo:hi]))
        S, sc = eng.estep(c, b, W)
        lab, G = eng.gibbs_labels(c, b, W, seed=3, sweep=1)
        R = eng.random_resp_stats(K, seed=2)
        q.put((rank, S.n.copy(), S.gating_counts.copy(), G.n.copy(), G.gating_counts.copy(), sc[0], R.gating_counts.copy()))
    finally:
        dist.destroy_process_group()


def test_two_rank_nan_rows_on_one_rank_only():
    """Rows with NaN on one shard only: the gating counts every rank ends up with — statistics of the complete
    rows + the share of the NaN rows — equal the single-process engine's, for the softmax pass, the label pass and the
    random start; the ranks that hold no such rows still join the extra sum."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_nan, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from oracle_engine import OracleEngine
    rng = np.random.default_rng(5)
    N, D, K = 400, 3, 5
    X = rng.standard_normal((N, D)) * 2.
    X[[210, 333, 399], 1] = np.nan
    A = rng.standard_normal((K, D, D)); W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D)
    b, c = rng.standard_normal((K, D)), rng.standard_normal(K)
    eng = OracleEngine(); eng.upload(X)
    S, sc = eng.estep(c, b, W)
    lab, G = eng.gibbs_labels(c, b, W, seed=3, sweep=1)
    assert S.n_rows is not None and abs(S.gating_counts.sum() - N) < 1e-9 and G.gating_counts.sum() == N
    for r in range(2):
        assert np.allclose(res[r][1], S.n, rtol=1e-12) and np.allclose(res[r][2], S.gating_counts, rtol=1e-12)
        assert np.array_equal(res[r][3], G.n) and np.array_equal(res[r][4], G.gating_counts)
        assert abs(res[r][5] - sc[0]) < 1e-9 * abs(sc[0])
        assert abs(res[r][6].sum() - N) < 1e-9
    assert np.array_equal(res[0][2], res[1][2])
