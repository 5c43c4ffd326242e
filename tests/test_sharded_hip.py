"""GPU (-m gpu): world size 2 of the REAL HIP engines on one GPU — two fresh interpreters (tests/sharded_hip_worker.py), both on
cuda:0, torch.distributed over gloo, each `ShardedEngine(HipEngine(0), row_offset=...)` on its shard — against the single-process
HIP run of the same problem (reference: one process, mimo/mixtures/gmm.py:207-237,261-287):
Philox labels bit for bit (global-row counters), statistics / bound traces / posteriors to 1e-12, and the rank-ordered sum bit for
bit equal to adding the two single-process blocks in rank order; NaN rows on one rank only; a rank whose numpy.random state differs
is caught before the labels diverge.  (CPU counterpart with the oracle-backed double: tests/test_sharded_gloo.py; one-rank RCCL
routes: test_gpu_parity.py::test_sharded_device_route_over_one_rank_rccl.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden, rel_err
import model_checks as mc
from sharded_hip_worker import random_case, nan_rows_of

pytestmark = pytest.mark.gpu


def run_ranks(case, tmp_path, world=2):
    """Start `world` worker interpreters (children of this process; they run side by side with this process's own context on
    the card: world + 1 <= 6 GPU processes) and return their result dictionaries in rank order."""
    port = 36000 + (os.getpid() * 7 + abs(hash(case))) % 3000
    outs = [str(tmp_path / f"rank{r}.npz") for r in range(world)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sharded_hip_worker.py"), case, str(r), str(world),
                               str(port), outs[r]], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=600)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} of '{case}' failed:\n" + logs[r][-4000:]
    return [dict(np.load(o, allow_pickle=False)) for o in outs]


@pytest.mark.parametrize("name", ["gmm_c2_d16_k16_n4099", "gmm_c3_d8_k32_n4099"])
def test_two_hip_ranks_gmm_fixture(engine, tmp_path, name):
    """Reference fixture on two HIP ranks: the 20-iteration bound trace and the posterior equal the reference's single-process
    values and the single-process HIP run; Philox labels of the shards concatenate to the unsharded draw."""
    res = run_ranks("gmm:" + name, tmp_path)
    g = load_golden(name)
    assert np.array_equal(res[0]["vlb"], res[1]["vlb"]) and np.array_equal(res[0]["S"], res[1]["S"])
    assert rel_err(res[0]["vlb"], g["vi_vlb"]) < 1e-8
    for i, ref in enumerate(mc.nw_of(g, "vi_post")):
        assert np.array_equal(res[0][f"post{i}"], res[1][f"post{i}"]) and rel_err(res[0][f"post{i}"], ref) < 1e-6
    import numpy.random as npr
    kind, model = mc.build_gmm(g, engine)
    mc.load_gmm_state(model, g, kind)
    eng = model._bind(g["X"])
    lab1, G1 = eng.gibbs_labels(*model.likelihood.canonical(), seed=1337, sweep=3)
    assert np.array_equal(np.concatenate([res[0]["labels"], res[1]["labels"]]), lab1)
    assert np.array_equal(lab1, g["labels_philox"])
    assert np.array_equal(res[0]["Gn"], G1.n) and rel_err(res[0]["Gsxx"], G1.sxx) < 1e-12
    npr.seed(77)
    vlb1 = model.meanfield_coordinate_descent(g["X"], randomize=False, maxiter=len(g["vi_vlb"]), tol=0., progress_bar=False)
    assert rel_err(res[0]["vlb"], np.array(vlb1)) < 1e-11
    for i, p in enumerate(model.components.posterior.params):
        assert rel_err(res[0][f"post{i}"], p) < 1e-10
    S1, sc1 = eng.estep(*model.canonical_expected())
    assert rel_err(res[0]["S"], S1.packed()) < 1e-10 and abs(res[0]["sc"][0] - sc1[0]) < 1e-10 * abs(sc1[0])


def test_two_hip_ranks_ilr_fixture(engine, tmp_path):
    """The reference's ILR default shape (50 experts over dx = dy = 1: the narrow kernels) on two HIP ranks."""
    name = "ilr_dx1_dy1_k50_n4099"
    res = run_ranks("ilr:" + name, tmp_path)
    g = load_golden(name)
    assert str(res[0]["softmax_kind"]) == "narrow" and str(res[0]["label_kind"]) == "narrow"
    assert np.array_equal(res[0]["vlb"], res[1]["vlb"])
    assert rel_err(res[0]["vlb"], g["vi_vlb"]) < 1e-7
    for i, ref in enumerate(mc.mnw_of(g, "vi_mpost")):
        assert rel_err(res[0][f"post{i}"], ref) < 1e-5
    assert np.array_equal(np.concatenate([res[0]["labels"], res[1]["labels"]]), g["labels_philox"])
    import numpy.random as npr
    kind, ilr = mc.build_ilr(g, engine)
    mc.load_ilr_state(ilr, g, kind)
    npr.seed(77)
    vlb1 = ilr.meanfield_coordinate_descent(g["X"], g["Y"], randomize=False, maxiter=len(g["vi_vlb"]), tol=0., progress_bar=False)
    assert rel_err(res[0]["vlb"], np.array(vlb1)) < 1e-11
    for i, p in enumerate(ilr.models.posterior.params):
        assert rel_err(res[0][f"post{i}"], p) < 1e-9


# one shape per kernel family (softmax pass / label pass): fused tile, row-owner, mid (softmax and label mode), two-stage (wide), narrow,
# small, streamed label kernel + one-pass label statistics
FAMILIES = [(16, 64, "fused", "rowwave"), (8, 256, "fused", "rowwave"), (20, 80, "mid", "rowwave"), (2, 50, "narrow", "narrow"),
            (2, 4, "small", "small"), (8, 32, "rowwave-vi", "rowwave"), (12, 6, "narrow", "narrow"), (24, 128, "two-stage", "rowwave"),
            (20, 16, "mid", "mid")]


@pytest.mark.parametrize("D,K,softmax_kind,label_kind", FAMILIES)
def test_two_hip_ranks_per_kernel_family(engine, tmp_path, D, K, softmax_kind, label_kind):
    """2 x 70 001 random rows per kernel family: everything the sharded engine returns equals the single-process HIP run —
    labels (Philox and host uniforms) bit for bit, counts exactly, statistics to 1e-12 —, a second pass returns the same bits,
    and the sum over the ranks IS block 0 + block 1 of the single-process runs on the two shards, bit for bit."""
    per = 70001
    res = run_ranks(f"random:{D}:{K}:{per}", tmp_path)
    N = 2 * per
    Z, c, b, W = random_case(D, K, N)
    assert str(res[0]["softmax_kind"]) == softmax_kind and str(res[0]["label_kind"]) == label_kind
    for key in ("S", "sc", "G", "Gu", "L", "R"):
        assert np.array_equal(res[0][key], res[1][key], equal_nan=True), key  # every rank holds the same global block (sc[1:] are NaN without the entropy split)
    assert np.array_equal(res[0]["S"], res[0]["S2"]) and res[0]["sc"][0] == res[0]["sc2"][0]
    engine.upload(Z)
    S1, sc1 = engine.estep(c, b, W)
    lab1, G1 = engine.gibbs_labels(c, b, W, seed=21, sweep=5)
    u = np.random.default_rng(9).random(N)
    labu1, Gu1 = engine.gibbs_labels(c, b, W, u=u)
    assert np.array_equal(np.concatenate([res[0]["labels"], res[1]["labels"]]), lab1)
    assert np.array_equal(np.concatenate([res[0]["labels_u"], res[1]["labels_u"]]), labu1)
    Kc = 1 + D + D * D
    G, Gu = res[0]["G"].reshape(K, Kc), res[0]["Gu"].reshape(K, Kc)
    assert np.array_equal(G[:, 0], np.bincount(lab1, minlength=K)) and np.array_equal(Gu[:, 0], np.bincount(labu1, minlength=K))
    assert rel_err(G, G1.packed()) < 1e-12 and rel_err(Gu, Gu1.packed()) < 1e-12
    assert rel_err(res[0]["S"], S1.packed()) < 1e-12 and abs(res[0]["sc"][0] - sc1[0]) < 1e-12 * abs(sc1[0])
    assert rel_err(res[0]["L"], G1.packed()) < 1e-12
    assert abs(res[0]["R"].reshape(K, Kc)[:, 0].sum() - N) < 1e-9 * N
    # rank order: the shards' single-process blocks, added block 0 first
    engine.upload(np.ascontiguousarray(Z[:per]))
    Sa, sca = engine.estep(c, b, W)
    _, Ga = engine.gibbs_labels(c, b, W, seed=21, sweep=5)
    engine.upload(np.ascontiguousarray(Z[per:])); engine.set_row_offset(per)
    Sb, scb = engine.estep(c, b, W)
    _, Gb = engine.gibbs_labels(c, b, W, seed=21, sweep=5)
    engine.set_row_offset(0)
    assert np.array_equal(res[0]["S"], (Sa.packed() + Sb.packed())) and res[0]["sc"][0] == sca[0] + scb[0]
    assert np.array_equal(res[0]["G"].reshape(K, Kc), Ga.packed() + Gb.packed())


@pytest.mark.parametrize("D,K", [(16, 64), (2, 50), (20, 80)])
def test_two_hip_ranks_nan_rows_on_one_rank(engine, tmp_path, D, K):
    """Rows with NaN on the second rank only: statistics without the rows, gating counts with their share (softmax, label and
    random-start passes), sum_n lse_n with them — equal to the single-process HIP engine on the same data."""
    per = 70001
    res = run_ranks(f"nan:{D}:{K}:{per}", tmp_path)
    N = 2 * per
    Z, c, b, W = random_case(D, K, N)
    Z[nan_rows_of(N), 0] = np.nan
    assert int(res[0]["n_bad"]) == 0 and int(res[1]["n_bad"]) == len(nan_rows_of(N))
    engine.upload(Z)
    assert engine.n_bad == len(nan_rows_of(N))
    S1, sc1 = engine.estep(c, b, W)
    lab1, G1 = engine.gibbs_labels(c, b, W, seed=21, sweep=5)
    for r in range(2):
        assert rel_err(res[r]["S"], S1.packed()) < 1e-12 and abs(res[r]["sc"][0] - sc1[0]) < 1e-12 * abs(sc1[0])
        assert rel_err(res[r]["Sg"], S1.gating_counts) < 1e-12 and abs(res[r]["Sg"].sum() - N) < 1e-9 * N
        assert np.array_equal(res[r]["Gg"], G1.gating_counts) and res[r]["Gg"].sum() == N
        assert rel_err(res[r]["G"], G1.packed()) < 1e-12
        assert abs(res[r]["Rg"].sum() - N) < 1e-9 * N
    assert np.array_equal(np.concatenate([res[0]["labels"], res[1]["labels"]]), lab1)
    engine.upload(np.zeros((4, D)))


def test_two_hip_ranks_divergent_host_generators_are_caught(tmp_path):
    """A rank that seeds numpy.random differently draws other component blocks in its Gibbs sweep: `check_replicated_once` raises
    on every rank at the first label pass that uses drawn blocks (here the second: the run starts from the given posterior); with
    identical seeds the same run goes through and its labels are those of ... each other."""
    res = run_ranks("diverge", tmp_path)
    assert bool(res[0]["raised"]) and bool(res[1]["raised"]), (str(res[0]["msg"]), str(res[1]["msg"]))
    assert len(res[0]["labels_ok"]) + len(res[1]["labels_ok"]) == len(load_golden("gmm_c3_d8_k32_n4099")["X"])
