"""One rank of a world-size-N run of the REAL HIP engine under ShardedEngine, all ranks on cuda:0, torch.distributed over
gloo (the host route of mimo_amd/sharded.py: _allreduce_host).  Started by tests/test_sharded_hip.py as a fresh interpreter per
rank (never an exec from a process that has touched the GPU); writes what it computed to an .npz the parent compares with the
single-process HIP run.

    python tests/sharded_hip_worker.py CASE RANK WORLD PORT OUT.npz

CASE: gmm:<fixture> | ilr:<fixture> | random:<Dz>:<K>[:<rows per rank>] | nan:<Dz>:<K> | diverge
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def random_case(D, K, N):
    """The same problem in every process (the parent rebuilds it for the single-process run)."""
    rng = np.random.default_rng(4200 + 100 * D + K)
    Z = rng.standard_normal((N, D)) * 2.0 + rng.standard_normal(D)
    A = rng.standard_normal((K, D, D))
    W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D)
    mu = rng.standard_normal((K, D)) * 2
    b = np.einsum('kde,ke->kd', W, mu)
    c = -0.5 * np.einsum('kd,kd->k', mu, b) + rng.standard_normal(K) * 0.1
    return Z, c, b, W


def nan_rows_of(N):
    """Rows that hold a NaN in the `nan` case: all in the second half (rank 1 of 2)."""
    return np.array([N // 2 + 3, N // 2 + 4097, N - 1, N - 70])


def main():
    case, rank, world, port, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    import numpy.random as npr
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    try:
        from conftest import load_golden
        from mimo_amd.engine import HipEngine
        from mimo_amd.sharded import ShardedEngine, shard_rows
        import model_checks as mc
        kind, *rest = case.split(":")
        inner = HipEngine(0)
        if kind in ("gmm", "ilr"):
            g = load_golden(rest[0])
            N = len(g["X"])
            lo, hi = shard_rows(N, rank, world)
            eng = ShardedEngine(inner, row_offset=lo)
            npr.seed(77)                         # identical host streams on every rank (the VI driver draws point estimates)
            if kind == "gmm":
                gk, model = mc.build_gmm(g, eng)
                mc.load_gmm_state(model, g, gk)
                Xl = np.ascontiguousarray(g["X"][lo:hi])
                model._bind(Xl)
                labels, G = eng.gibbs_labels(*model.likelihood.canonical(), seed=1337, sweep=3)
                vlb = model.meanfield_coordinate_descent(Xl, randomize=False, maxiter=len(g["vi_vlb"]), tol=0., progress_bar=False)
                post = model.components.posterior.params
                S, sc = eng.estep(*model.canonical_expected())
            else:
                gk, model = mc.build_ilr(g, eng)
                mc.load_ilr_state(model, g, gk)
                Xl, Yl = np.ascontiguousarray(g["X"][lo:hi]), np.ascontiguousarray(g["Y"][lo:hi])
                model._bind(Xl, Yl)
                labels, G = eng.gibbs_labels(*model.likelihood.canonical(), seed=1337, sweep=3)
                vlb = model.meanfield_coordinate_descent(Xl, Yl, randomize=False, maxiter=len(g["vi_vlb"]), tol=0., progress_bar=False)
                post = model.models.posterior.params
                S, sc = eng.estep(*model.canonical_expected())
            res = dict(lo=lo, labels=labels, Gn=G.n, Gsxx=G.sxx, vlb=np.array(vlb), S=S.packed(), sc=sc,
                       softmax_kind=inner.plan(model.size)["kind"], label_kind=inner.plan(model.size, gibbs=True)["kind"],
                       **{f"post{i}": np.asarray(p) for i, p in enumerate(post)})
        elif kind in ("random", "nan"):
            D, K = int(rest[0]), int(rest[1])
            per = int(rest[2]) if len(rest) > 2 else 70001
            N = world * per
            Z, c, b, W = random_case(D, K, N)
            if kind == "nan":
                Z[nan_rows_of(N), 0] = np.nan
            lo, hi = shard_rows(N, rank, world)
            eng = ShardedEngine(inner, row_offset=lo)
            eng.upload(np.ascontiguousarray(Z[lo:hi]))
            S, sc = eng.estep(c, b, W)
            S2, sc2 = eng.estep(c, b, W)
            labels, G = eng.gibbs_labels(c, b, W, seed=21, sweep=5)
            u = np.random.default_rng(9).random(N)[lo:hi]
            labels_u, Gu = eng.gibbs_labels(c, b, W, u=u)
            L = eng.label_stats(labels, K)
            R = eng.random_resp_stats(K, seed=3)
            res = dict(lo=lo, S=S.packed(), sc=sc, S2=S2.packed(), sc2=sc2, labels=labels, G=G.packed(), labels_u=labels_u, Gu=Gu.packed(),
                       L=L.packed(), R=R.packed(), n_bad=inner.n_bad, Sg=S.gating_counts, Gg=G.gating_counts, Rg=R.gating_counts,
                       softmax_kind=inner.plan(K)["kind"], label_kind=inner.plan(K, gibbs=True)["kind"])
        elif kind == "diverge":
            # a Gibbs run whose ranks seed numpy.random differently: the first label pass must raise on every rank
            g = load_golden("gmm_c3_d8_k32_n4099")
            N = len(g["X"])
            lo, hi = shard_rows(N, rank, world)
            eng = ShardedEngine(inner, row_offset=lo)
            gk, model = mc.build_gmm(g, eng)
            mc.load_gmm_state(model, g, gk)
            npr.seed(500 + rank)
            raised, msg = False, ""
            try:
                model.resample(np.ascontiguousarray(g["X"][lo:hi]), init_labels='posterior', maxiter=2, progress_bar=False,
                               label_rng='philox', seed=5)
            except RuntimeError as exc:
                raised, msg = "differ between the ranks" in str(exc), str(exc)
            res = dict(raised=raised, msg=np.array(msg))
            # ... and the same run with identical seeds goes through
            eng2 = ShardedEngine(inner, row_offset=lo)
            gk, model = mc.build_gmm(g, eng2)
            mc.load_gmm_state(model, g, gk)
            npr.seed(500)
            model.resample(np.ascontiguousarray(g["X"][lo:hi]), init_labels='posterior', maxiter=2, progress_bar=False,
                           label_rng='philox', seed=5)
            res["labels_ok"] = model.labels_
            res["lo"] = lo
        else:
            raise ValueError(case)
        np.savez(out, **res)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
