import sys, os, time, cProfile, pstats
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import bench
from mimo_amd.engine import HipEngine
eng = HipEngine(0)
N = 4_000_000
for name in ("sine", "c4"):
    cfg = bench.CONFIGS[name]
    D, K = cfg[2], cfg[3]
    dx, dy = (8, 4) if D == 12 else (D // 2, D - D // 2)
    model = bench.build_model(cfg, eng)
    rng = np.random.default_rng(0)
    X = rng.standard_normal((N, dx)); Y = rng.standard_normal((N, dy))
    model.meanfield_coordinate_descent(X[:20000], Y[:20000], maxiter=3, progress_bar=False)
    for it in range(2): model.meanfield_prediction(X)
    ts = []
    for it in range(5):
        t0 = time.perf_counter(); out = model.meanfield_prediction(X); ts.append(time.perf_counter() - t0)
    print(name, "meanfield_prediction ms:", [round(t * 1e3, 1) for t in ts])
    pr = cProfile.Profile(); pr.enable(); model.meanfield_prediction(X); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
