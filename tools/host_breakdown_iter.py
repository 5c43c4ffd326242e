"""Diagnostic: where one iteration of the PUBLIC mean-field loop (meanfield_iteration: what bench.py times) spends its wall time.
    python tools/host_breakdown_iter.py c1|c2|c4 [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mimo_amd.engine import HipEngine
name = sys.argv[1] if len(sys.argv) > 1 else "c1"
cfg = bench.CONFIGS[name]; desc, N, D, K, mode = cfg
if len(sys.argv) > 2: N = int(float(sys.argv[2]))
X = bench.make_data(N, D, K, seed=1337, device="cuda:0", ilr=(mode == "ilr")); torch.cuda.synchronize()
hip = HipEngine(0); hip.upload(X)
model = bench.build_model(cfg, hip)
S = hip.label_stats(np.random.default_rng(1).integers(0, K, size=N).astype(np.int32), K)
names = ("update", "canonical", "launch", "refresh (rvs)", "prior_terms", "wait")
T = dict.fromkeys(names, 0.0)
reps = 200
for it in range(reps + 20):
    if it == 20:
        T = dict.fromkeys(names, 0.0); t_all = time.perf_counter()
    t0 = time.perf_counter(); model._update_from_stats(S, sample=False)
    t1 = time.perf_counter(); th = model.canonical_expected()
    t2 = time.perf_counter(); hip.estep_async(*th)
    t3 = time.perf_counter(); model._refresh_likelihoods()
    t4 = time.perf_counter(); pt = model._vlb_prior_terms()
    t5 = time.perf_counter(); S, sc = hip.estep_wait()
    t6 = time.perf_counter()
    for k, v in zip(names, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)): T[k] += v
tot = (time.perf_counter() - t_all) / reps
print(name, f"direct_out={os.environ.get('MIMO_DIRECT_OUT', '1')}", "per iteration [us]:", {k: round(v / reps * 1e6, 1) for k, v in T.items()}, "total", round(tot * 1e6, 1))
