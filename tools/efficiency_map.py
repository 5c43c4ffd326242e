"""Diagnostic: one VI pass and one Gibbs sweep over a grid of (Dz, K) against BOTH roofs of SURVEY.md section 8(d) — device time of all
kernels of the pass (HIP events):
   f64 : algorithmic flops / t / 78.6 TFLOP/s        (F_E = D(D+1)+3D+8 per evaluation; F_S = (D+1)(D+2)+1 per evaluation (VI) / per datum (Gibbs))
   hbm : algorithmic bytes / t / 8 TB/s              (VI: 8 N Dz — the data once; Gibbs: + 4 N labels written, and where the statistics
                                                       are a second pass (mimo_plan) the data and the labels once more)
   max : the binding one — what SURVEY asks to be reported; a low-intensity cell (few components over narrow rows) is HBM work.
    python tools/efficiency_map.py [N] [vi|gibbs|both]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd.engine import HipEngine
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
which = sys.argv[2] if len(sys.argv) > 2 else "both"
Ds = (2, 4, 8, 12, 16, 20, 24, 28, 32)
Ks = (4, 8, 16, 32, 48, 64, 96, 128, 192, 256)
eng = HipEngine(0)
rng = np.random.default_rng(0)
for mode in ("vi", "gibbs"):
    if which not in (mode, "both"):
        continue
    rows = {"f64": [], "hbm": [], "max": []}
    for D in Ds:
        Z = rng.standard_normal((N, D)); eng.upload(Z)
        line = {"f64": [], "hbm": [], "max": []}
        for K in Ks:
            A = rng.standard_normal((K, D, D)); W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D)
            b = rng.standard_normal((K, D)); c = rng.standard_normal(K)
            run = (lambda it: eng.gibbs_labels(c, b, W, seed=1, sweep=it, return_labels=False)) if mode == "gibbs" else (lambda it: eng.estep(c, b, W))
            for it in range(2): run(it)
            eng.profile(True); eng.profile_read(reset=True)
            for it in range(4): run(it)
            ms, n = eng.profile_read(reset=True); eng.profile(False)
            t = ms / n * 1e-3
            FE, FS = D * (D + 1) + 3 * D + 8, (D + 1) * (D + 2) + 1
            flops = N * K * FE + (N * K * FS if mode == "vi" else N * FS)
            plan = eng.plan(K, gibbs=(mode == "gibbs"))
            nbytes = 8.0 * N * D * (1 if mode == "vi" else max(1, min(2, plan["data_passes"]))) \
                + (4.0 * N * (1 if plan["data_passes"] <= 1 else 2) if mode == "gibbs" else 0.0)
            f, h = flops / t / 78.6e12, nbytes / t / 8e12
            line["f64"].append(f); line["hbm"].append(h); line["max"].append(max(f, h))
        for k in rows:
            rows[k].append(line[k])
        print(f"  [{mode}] Dz={D:2d} max: " + " ".join(f"{v:5.2f}" for v in line["max"]), flush=True)
    for k, title in (("f64", "fraction of 78.6 TFLOP/s (algorithmic flops)"), ("hbm", "fraction of 8 TB/s (algorithmic bytes)"),
                     ("max", "the binding roof: max of the two")):
        print(f"{mode}: {title} (N = {N}); rows Dz, columns K = {Ks}")
        for D, line in zip(Ds, rows[k]):
            print(f"  Dz={D:2d} " + " ".join(f"{v:5.2f}" for v in line))
