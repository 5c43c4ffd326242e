#!/usr/bin/env python3
"""bench.py — E-step datapoint-component evaluations per second (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5|c1]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no RANK / WORLD_SIZE in the environment launches the N ranks itself (fresh child
processes through torch.distributed.run on 127.0.0.1, before this process touches the GPU) and relays rank 0's JSON
line, so both forms of the driver's command work.

Workload (default c2 = BASELINE.json configs[1]): mean-field VB of a full-covariance GMM, N = 1e7 rows
PER GPU (weak scaling), D = 16, K = 64, synthetic float64 data already resident in HBM.  One timed
"step" is one iteration of the PUBLIC driver loop (`BayesianMixtureOfGaussians.meanfield_iteration`, the body of
`meanfield_coordinate_descent` with its defaults — including the reference's per-iteration
likelihood.params = posterior.rvs()): host conjugate update of all K posteriors, the fused HIP pass over the data
(log-densities -> softmax -> sufficient statistics -> ELBO terms; the (K,N) responsibilities never leave the GPU),
the RCCL all-reduce of the statistic block when N > 1, the point-estimate draws and the ELBO's prior terms on the
host while the kernel runs.  value = (rows on all ranks x K) / max-over-ranks step time.

The JSON line also carries
  roofline     : the dominant kernel against the float64 matrix-core peak (HIP events around every launch
                 on the launch stream, averaged over the timed steps; algorithmic flops of SURVEY.md §8(d)),
                 or against the HBM roof for the small-shape kernel (config c1)
  cpu_baseline : the NumPy restatement of the reference algorithm (oracle/, verified equal to the
                 reference on golden vectors) timed on this box's host cores on a bounded row sample
  sustained    : the same step repeated for >= 3 s after the timed steps (median / min ms per step).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

F64_MATRIX_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix (= vector) peak, vendor; tools/f64_rates measures 78.0
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)

CONFIGS = {
    # name: (description, N per GPU, D (Dz), K, mode)
    "c1": ("C1 shape at scale: the reference's own toy workload (examples/gmm/toy: 2-D, K=4 full-covariance GMM, "
           "mean-field VB), N=1e7 per GPU — small-shape VALU kernel, HBM-bound", 10_000_000, 2, 4, "vi"),
    "c2": ("C2: mean-field VB GMM, N=1e7 per GPU, D=16, K=64, full covariance", 10_000_000, 16, 64, "vi"),
    "c3": ("C3: DP-GMM truncated stick-breaking Kmax=256, N=1e7 per GPU, D=8, Gibbs (Philox labels)",
           10_000_000, 8, 256, "gibbs"),
    "c4": ("C4: mixture of linear-Gaussian experts, N=5e6 per GPU, x in R^8 -> y in R^4, K=64, mean-field VB",
           5_000_000, 12, 64, "ilr"),
    "c5": ("C5: mean-field VB GMM, N=1e7 per GPU, D=32, K=128 (two-stage path: E-step writing the responsibility table + statistics per column group)",
           10_000_000, 32, 128, "vi"),
    # not a BASELINE config: the shape the reference's own ILR examples default to (examples/ilr/evaluate_sine.py:35: 50 experts over
    # dx = dy = 1, i.e. Dz = 2) at scale — the narrow kernels on the 4x4x4 matrix instruction
    "sine": ("ILR defaults of the reference's examples at scale: mixture of 50 linear-Gaussian experts, x in R -> y in R, N=1e7 per GPU, "
             "mean-field VB", 10_000_000, 2, 50, "ilr"),
}


def algorithmic_flops_per_eval(D, mode):
    """F_E = D(D+1) + 3D + 8 (E-step), F_S = (D+1)(D+2) + 1 (statistics, per
    (datum, component) for VI; once per datum for Gibbs)."""
    FE = D * (D + 1) + 3 * D + 8
    FS = (D + 1) * (D + 2) + 1
    return FE, FS


def make_data(N, D, K, seed, device, ilr=False):
    """Synthetic mixture data generated on the GPU: K_true = min(K, 32) centres
    ~ N(0, 6^2 I), per-centre covariance A A'/D + 0.1 I, rows assigned uniformly."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    Kt = min(K, 32)
    hg = np.random.Generator(np.random.Philox(1337))          # identical centres on every rank
    dx, dyo = ((8, 4) if D == 12 else (D // 2, D - D // 2)) if ilr else (D, 0)
    centres = torch.tensor(hg.normal(0.0, 6.0, size=(Kt, dx)), device=device)
    A = hg.normal(size=(Kt, dx, dx))
    cov = A @ A.transpose(0, 2, 1) / dx + 0.1 * np.eye(dx)
    L = torch.tensor(np.linalg.cholesky(cov), device=device)
    z = torch.randint(Kt, (N,), generator=g, device=device)
    X = torch.empty((N, D), dtype=torch.float64, device=device)
    chunk = 1_000_000
    if ilr:
        Ak = torch.tensor(hg.normal(size=(Kt, dyo, dx)) / np.sqrt(dx), device=device)
        ck = torch.tensor(hg.normal(size=(Kt, dyo)), device=device)
    for s in range(0, N, chunk):
        zz = z[s:s + chunk]
        eps = torch.randn((zz.numel(), dx), dtype=torch.float64, generator=g, device=device)
        x = centres[zz] + torch.einsum('nde,ne->nd', L[zz], eps)
        X[s:s + chunk, :dx] = x
        if ilr:
            y = torch.einsum('nde,ne->nd', Ak[zz], x) + ck[zz] \
                + 0.3 * torch.randn((zz.numel(), dyo), dtype=torch.float64, generator=g, device=device)
            X[s:s + chunk, dx:] = y
    return X


def build_model(cfg, engine):
    from mimo_amd.distributions import (Dirichlet, TruncatedStickBreaking, CategoricalWithDirichlet,
                                        CategoricalWithStickBreaking, StackedNormalWisharts,
                                        StackedGaussiansWithNormalWisharts, StackedMatrixNormalWisharts,
                                        StackedLinearGaussiansWithMatrixNormalWisharts)
    from mimo_amd.mixtures import BayesianMixtureOfGaussians, BayesianMixtureOfLinearGaussians
    _, _, D, K, mode = cfg
    np.random.seed(1337)      # identical host state on every rank
    if mode == "gibbs":
        gating = CategoricalWithStickBreaking(K, TruncatedStickBreaking(K, np.ones(K), 5. * np.ones(K)))
    else:
        gating = CategoricalWithDirichlet(K, Dirichlet(K, np.ones(K)))
    if mode == "ilr":
        dx, dy = (8, 4) if D == 12 else (D // 2, D - D // 2)
        bprior = StackedNormalWisharts(K, dx, np.zeros((K, dx)), 1e-2 * np.ones(K),
                                       np.stack(K * [1e2 * np.eye(dx)]), (dx + 1.) * np.ones(K) + 1e-16)
        basis = StackedGaussiansWithNormalWisharts(K, dx, bprior, engine=engine)
        mprior = StackedMatrixNormalWisharts(K, dx + 1, dy, np.zeros((K, dy, dx + 1)),
                                             np.stack(K * [1e-2 * np.eye(dx + 1)]),
                                             np.stack(K * [np.eye(dy)]), (dy + 1.) * np.ones(K) + 1e-16)
        models = StackedLinearGaussiansWithMatrixNormalWisharts(K, dx + 1, dy, mprior, engine=engine)
        return BayesianMixtureOfLinearGaussians(K, dx, dy, gating, basis, models, engine=engine)
    prior = StackedNormalWisharts(K, D, np.zeros((K, D)), 1e-2 * np.ones(K), np.stack(K * [np.eye(D)]),
                                  (D + 1.) * np.ones(K) + 1e-8)
    comps = StackedGaussiansWithNormalWisharts(K, D, prior, engine=engine)
    return BayesianMixtureOfGaussians(gating, comps, engine=engine)


def _blas_threads():
    threads = os.cpu_count()
    try:
        from threadpoolctl import threadpool_info
        blas = [p for p in threadpool_info() if p.get("user_api") == "blas"]
        if blas:
            threads = blas[0]["num_threads"]
    except Exception:
        pass
    return int(threads)


def _timed_rows(sweep, X_host, probe_rows, budget_s, max_rows):
    """Rows that take about `budget_s` seconds of `sweep` (linear in the rows: no cross-datum coupling), and their time."""
    sweep(X_host[:min(len(X_host), probe_rows // 4)])                      # warm the BLAS threads / caches
    t0 = time.time(); sweep(X_host[:probe_rows]); dt = time.time() - t0
    rows = int(min(len(X_host), max_rows, max(probe_rows, 1024 * round(probe_rows * budget_s / max(dt, 1e-4) / 1024))))
    t0 = time.time(); sweep(X_host[:rows]); dt = time.time() - t0
    return rows, dt


def cpu_baseline(cfg, X_host):
    """The NumPy restatement of the reference's sweep for this configuration (oracle/, verified equal to the reference on
    the golden vectors), timed on the host cores on a bounded prefix of the same data:
      vi    — E-step table in 1024-row chunks (the reference's (K,N,D,D) replication forces them: gaussian.py:481-485), softmax,
              weighted_statistics;
      gibbs — the label step un-chunked (gaussian.py:510-521 table + mimo/utils/stats.py:8-21 draw) + one_hot +
              weighted_statistics of the drawn labels (gaussian.py:491-502), rows bounded by the (K,N,D) temporaries (2 GB);
      ilr   — the VI sweep of the linear-Gaussian experts (bayesian.py:933-947 + 287-301 tables in chunks, softmax,
              lingauss.py:306-322 + gaussian.py:491-502 statistics)."""
    from oracle import mimo_oracle as O
    _, _, D, K, mode = cfg
    rng = np.random.default_rng(0)
    if mode == "gibbs":
        A = rng.standard_normal((K, D, D))
        mus, lmbdas = rng.standard_normal((K, D)) * 3, A @ A.transpose(0, 2, 1) / D + np.eye(D)
        probs = rng.dirichlet(np.ones(K))

        def sweep(x):
            lp = O.gmm_log_complete_likelihood(x, mus, lmbdas, probs)
            labels = O.sample_discrete_from_log(lp, rng.random((1, len(x))))
            return O.gauss_weighted_statistics(x, O.one_hot(labels, K))
        what = "one Gibbs label step + weighted_statistics of the drawn labels, un-chunked"
        probe, max_rows = 8192, int(2e9 / (8 * K * D))
    elif mode == "ilr":
        dx, dy = (8, 4) if D == 12 else (D // 2, D - D // 2)
        A = rng.standard_normal((K, dx, dx))
        bpost = (rng.standard_normal((K, dx)) * 3, np.full(K, 100.0), np.linalg.inv(A @ A.transpose(0, 2, 1) / dx + np.eye(dx)) / 50.,
                 np.full(K, 60.0))
        B = rng.standard_normal((K, dy, dy))
        mpost = (rng.standard_normal((K, dy, dx + 1)), np.stack(K * [np.eye(dx + 1)]) * 50.,
                 np.linalg.inv(B @ B.transpose(0, 2, 1) / dy + np.eye(dy)) / 50., np.full(K, 60.0))
        gpost = np.full(K, 50.0)

        def sweep(z):
            x, y = z[:, :dx], z[:, dx:]
            r = O.responsibilities(O.ilr_expected_log_complete_likelihood(x, y, bpost, mpost, 'dirichlet', gpost))
            return O.gauss_weighted_statistics(x, r), O.lingauss_weighted_statistics(x, y, r)
        what = "one VI sweep of the linear-Gaussian experts (E-step tables in chunks + softmax + both weighted_statistics)"
        probe, max_rows = 4096, len(X_host)
    else:
        A = rng.standard_normal((K, D, D))
        post = (rng.standard_normal((K, D)) * 3, np.full(K, 100.0), np.linalg.inv(A @ A.transpose(0, 2, 1) / D + np.eye(D)) / 50.,
                np.full(K, 60.0))
        gpost = np.full(K, 50.0)

        def sweep(x):
            ll = O.gmm_expected_log_complete_likelihood(x, post, 'dirichlet', gpost, chunk=1024)
            r = O.responsibilities(ll)
            return O.gauss_weighted_statistics(x, r)
        what = "one VI sweep (E-step in 1024-row chunks + softmax + weighted_statistics)"
        probe, max_rows = 8192, len(X_host)
    rows, dt = _timed_rows(sweep, X_host, probe, 15.0, max_rows)           # ~15 s of CPU work
    out = {"value": rows * K / dt, "unit": "evals/s", "cores": _blas_threads(), "kind": "port",
           "sample": f"first {rows} rows of the same synthetic data, {what} of the NumPy restatement of the reference, {dt:.1f} s"}
    try:      # per-core figure: the same sweep with the BLAS pool limited to one thread, ~5 s
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1, user_api="blas"):
            r1, d1 = _timed_rows(sweep, X_host, probe // 2, 5.0, max_rows)
        out["one_core"] = {"value": r1 * K / d1, "unit": "evals/s", "cores": 1, "sample": f"first {r1} rows, {d1:.1f} s"}
    except Exception:
        pass
    return out


def _rank_ordered(engine):
    """Which association the sum over the ranks has: the torch route follows ShardedEngine (MIMO_SHARDED_RANK_ORDER), the native
    route the library's own switch (mimo_comm.cpp: MIMO_COMM_RANK_ORDER, default on)."""
    if getattr(engine, "_native", False):
        return os.environ.get("MIMO_COMM_RANK_ORDER", "1") != "0"
    return bool(getattr(engine, "_rank_order", False))


def self_launch(args, argv, launcher=None):
    """`python bench.py --gpus N` (N > 1) outside a distributed launcher: start the N ranks as children of a fresh
    torch.distributed.run and relay rank 0's JSON line.  Nothing in this process has touched the GPU (torch is not
    even imported yet) and nothing is exec'ed: the launcher is a child process, its return code is ours."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), launcher or os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    js = [ln for ln in lines if ln.lstrip().startswith("{") and '"metric"' in ln]
    for ln in lines:
        if not js or ln is not js[-1]:
            print(ln, file=sys.stderr)
    if js:
        print(js[-1], flush=True)           # the JSON line is the last line of stdout
    if p.returncode != 0 or not js:
        raise SystemExit(p.returncode or 1)
    raise SystemExit(0)


def main(argv=None, engine_factory=None, launcher=None):
    """`engine_factory` / `launcher` are for callers that import this module (tests/bench_dry_run.py runs the step loop on the CPU
    with an engine double: the ranks then sit on gloo and the line is marked dry_run and carries no measurement); the script
    itself never sets them and imports nothing outside the product package (+ oracle/ for the cpu_baseline leg)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--rows", type=int, default=0, help="override rows per GPU (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sustained", action="store_true")
    ap.add_argument("--no-sample", action="store_true",
                    help="drop the reference's per-iteration likelihood.params = posterior.rvs() (sample_likelihood=False)")
    ap.add_argument("--sustained-seconds", type=float, default=10.0,
                    help="length of the sustained leg after the timed steps (>= 10 s: a 5-second utilisation sampler must see it)")
    args = ap.parse_args(argv)

    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None and "RANK" not in os.environ:
        self_launch(args, sys.argv[1:] if argv is None else list(argv), launcher)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    dry = engine_factory is not None
    if not dry:
        torch.cuda.set_device(local_rank)
    device = "cpu" if dry else f"cuda:{local_rank}"
    dist = None
    force_dist = os.environ.get("MIMO_BENCH_FORCE_DIST") == "1"      # exercise the RCCL path with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))

    from mimo_amd.sharded import ShardedEngine

    cfg = CONFIGS[args.config]
    desc, N, D, K, mode = cfg
    if args.rows:
        N = args.rows
    X = make_data(N, D, K, seed=1337 + rank, device=device, ilr=(mode == "ilr"))
    if dry:
        hip = engine_factory()
        hip.upload(X.numpy())
    else:
        from mimo_amd.engine import HipEngine
        torch.cuda.synchronize()
        hip = HipEngine(local_rank)                    # (own stream; ShardedEngine moves it onto the stream of its all-reduce)
        hip.upload(X)                                  # borrows the device tensor (no copy)
    engine = ShardedEngine(hip, row_offset=rank * N) if dist is not None else hip
    if dist is not None:
        hip.set_row_offset(rank * N)
    model = build_model(cfg, engine)

    # initial posterior: one M-step from seeded random hard labels
    labels0 = np.random.default_rng(4242 + rank).integers(0, K, size=N).astype(np.int32)
    S = engine.label_stats(labels0, K)

    param_rng = np.random.Generator(np.random.Philox(99))     # identical on every rank
    sample = not args.no_sample

    def step(S, it):
        """One iteration of the public driver loop (mixtures/gmm.py, ilr.py)."""
        if mode == "gibbs":
            _, S2 = model.gibbs_iteration(engine, S, it, label_rng='philox', seed=2024, param_rng=param_rng,
                                          stats=True, return_labels=False)
            return S2, None
        return model.meanfield_iteration(engine, S, sample_likelihood=sample)

    def barrier():
        if not dry:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            if not dry:
                torch.cuda.synchronize()

    def prof(on):
        """(total kernel ms, passes, per-kernel breakdown) since the last call; resets the counters."""
        if hasattr(hip, "profile"):
            hip.profile(on)
            kin = hip.profile_kernels() if hasattr(hip, "profile_kernels") else {}
            ms, n = hip.profile_read(reset=True)
            return ms, n, kin
        return 0.0, 0, {}

    vlb = []
    for it in range(args.warmup):
        S, v = step(S, it)
    prof(True)
    barrier()
    t0 = time.perf_counter()
    for it in range(args.steps):
        S, v = step(S, args.warmup + it)
        vlb.append(v)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches, kinfo = prof(True)
    kinfo = {k: {"ms_per_launch": v["ms"] / max(v["launches"], 1), "launches_per_step": v["launches"] / max(args.steps, 1)}
             for k, v in kinfo.items()}

    rank_ms = [elapsed / args.steps * 1e3]
    if dist is not None:          # the slowest rank's time is the step time; every rank's own time goes into the line
        t = torch.zeros(world, dtype=torch.float64, device=device)
        t[rank] = elapsed
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        rank_ms = [float(v) / args.steps * 1e3 for v in t.cpu()]
        elapsed = float(t.max().item())

    sustained = None
    if not args.no_sustained and not dry and dist is None:
        # the same step for >= 10 s after the timed steps: a 20-launch burst says nothing about sustained FP64 clocks, and an
        # independent 5-second utilisation sampler has to see the load at least once
        ts, t_end, it = [], time.perf_counter() + max(args.sustained_seconds, 0.0), args.warmup + args.steps
        while time.perf_counter() < t_end or len(ts) < 5:
            torch.cuda.synchronize()
            a = time.perf_counter()
            S, _ = step(S, it)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - a) * 1e3)
            it += 1
        sustained = {"seconds": float(np.sum(ts)) / 1e3, "steps": len(ts), "ms_per_step_median": float(np.median(ts)),
                     "ms_per_step_min": float(np.min(ts)), "ms_per_step_last10_median": float(np.median(ts[-10:])),
                     # in-kernel counters (s_memtime / s_memrealtime) of a short float64 loop right after the leg
                     "shader_clock_mhz_under_f64_load": hip.shader_clock_mhz() if hasattr(hip, "shader_clock_mhz") else None}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * N * K / (elapsed / args.steps)
        FE, FS = algorithmic_flops_per_eval(D, mode)
        flops_per_launch = N * K * FE + (N * K * FS if mode != "gibbs" else N * FS)
        per_step = max(args.steps, 1)
        k_ms = kernel_ms / per_step            # device time of ALL kernels of the pass, per step (HIP events on the launch stream)
        achieved = flops_per_launch / (max(k_ms, 1e-9) * 1e-3) / 1e12
        plan = hip.plan(K, gibbs=(mode == "gibbs")) if hasattr(hip, "plan") else {}
        # algorithmic HBM bytes of one pass: the data once (+ labels written by a Gibbs pass and read back by an
        # unfused statistics pass); a two-stage plan also writes the (K, N) table and reads it once per column group
        data_bytes = 8.0 * N * D
        hbm_bytes = data_bytes * max(1, plan.get("data_passes", 1)) + (4.0 * N * plan.get("label_passes", 1) if mode == "gibbs" else 0.0)
        table_bytes = 8.0 * N * K * (1 + plan.get("table_reads", 0)) if plan.get("table_in_hbm") else 0.0
        out = {
            "metric": "E-step datapoint-component evals/sec",
            "value": None if dry else value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "n_ranks_seen": dist.get_world_size() if dist is not None else 1,
            "ms_per_step_per_rank": {"min": min(rank_ms), "max": max(rank_ms), "all": rank_ms},
            "allreduce_bytes_per_step": (8 * (K * (1 + D + D * D) + 4)) if dist is not None else 0,
            "allreduce_route": (("libmimo_hip RCCL communicator" if getattr(engine, "_native", False) else
                                 "torch.distributed (" + dist.get_backend() + ")")
                                + (", all-gather + sum in rank order" if _rank_ordered(engine) else ", all_reduce(sum)")
                                if dist is not None else None),
            "config": {"workload": desc, "rows_per_gpu": N, "Dz": D, "K": K,
                       "step": "one iteration of the public driver loop ("
                               + ("gibbs_iteration" if mode == "gibbs" else "meanfield_iteration, sample_likelihood=%s" % sample)
                               + "): host conjugate update of all K posteriors, HIP pass over the data (log-densities, "
                                 "softmax/label draw, sufficient statistics, ELBO terms)"
                               + (", RCCL all-reduce of the statistic block" if world > 1 else "")
                               + ", point-estimate draws + ELBO prior terms on the host while the kernel runs",
                       "plan": plan,
                       "resp_materialised_in_hbm": bool(plan.get("table_in_hbm", False)),
                       "parallelism": f"rows sharded over {world} GPU(s)"},
            "kernel_evals_per_s": N * K / (max(k_ms, 1e-9) * 1e-3),
        }
        hbm_block = {"algorithmic_bytes_per_step": hbm_bytes, "table_bytes_per_step": table_bytes,
                     "achieved_GBs": (hbm_bytes + table_bytes) / (max(k_ms, 1e-9) * 1e-3) / 1e9,
                     "frac_of_8TBs": (hbm_bytes + table_bytes) / (max(k_ms, 1e-9) * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if plan.get("kind") == "small":
            # small-shape VALU kernel: bound by HBM (8 N Dz bytes per pass), not by the matrix pipe
            out["roofline"] = {"bound": "hbm", "achieved": hbm_block["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": hbm_block["frac_of_8TBs"], "traffic": None, "kernel": "mimo::small_kernel",
                               "kernel_ms": k_ms, "launches": launches,
                               "algorithmic_bytes_per_launch": hbm_bytes,
                               "f64_valu": {"flops_per_eval": {"estep": FE, "stats": FS}, "achieved_TFLOPs": achieved,
                                            "frac_of_78.6": achieved / F64_MATRIX_PEAK_TFLOPS}}
        else:
            out["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": F64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": achieved / F64_MATRIX_PEAK_TFLOPS, "traffic": None,
                               "kernel": plan.get("kernel", "mimo::fused_kernel"), "kernel_ms": k_ms,
                               "launches": launches, "kernels_per_step": kinfo,
                               "flops_per_eval": {"estep": FE, "stats": FS}, "hbm": hbm_block}
        if dry:
            out["dry_run"] = True
            out["roofline"] = None
        try:     # HBM bytes per launch measured with rocprofv3 PMC counters for this workload (profiles/)
            with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
                t = json.load(f).get(args.config)
            if t and not args.rows and out["roofline"]:
                out["roofline"]["traffic"] = t["hbm_bytes_per_step"]
                out["roofline"]["traffic_source"] = t.get("source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, profiles/hbm_traffic.json")
        except Exception:
            pass
        if vlb and vlb[0] is not None:
            out["elbo_first_last"] = [float(vlb[0]), float(vlb[-1])]
        if sustained:
            out["sustained"] = sustained
        if world == 1 and not args.no_cpu_baseline and not dry:
            out["cpu_baseline"] = cpu_baseline(cfg, X[:600_000].cpu().numpy())
        else:
            out["cpu_baseline"] = None
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        try:                      # RCCL's banner sits in the C stdio buffer: drain it so the JSON line is last
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stderr.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
