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
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
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
