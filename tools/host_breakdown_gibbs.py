"""Diagnostic: where a Gibbs sweep's wall time goes beyond the kernels (C3 by default).  python tools/host_breakdown_gibbs.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mimo_amd.engine import HipEngine
from mimo_amd.mixtures.gmm import _component_stats
cfg = bench.CONFIGS["c3"]; desc, N, D, K, mode = cfg
if len(sys.argv) > 1: N = int(float(sys.argv[1]))
X = bench.make_data(N, D, K, seed=1337, device="cuda:0"); torch.cuda.synchronize()
hip = HipEngine(0); hip.upload(X)
model = bench.build_model(cfg, hip)
S = hip.label_stats(np.random.default_rng(1).integers(0, K, size=N).astype(np.int32), K)
rng = np.random.Generator(np.random.Philox(99))
T = {k: 0.0 for k in ("comp_resample", "gating_resample", "canonical", "label_pass")}
hip.profile(True)
for it in range(13):
    if it == 3:
        T = {k: 0.0 for k in T}; hip.profile_read(reset=True)
    t0 = time.perf_counter(); model.components.resample(None, stats=_component_stats(S, model.components), rng=rng)
    t1 = time.perf_counter(); model.gating.resample(None, counts=S.gating_counts)
    t2 = time.perf_counter(); th = model.likelihood.canonical()
    t3 = time.perf_counter(); _, S = hip.gibbs_labels(*th, seed=1, sweep=it, stats=True, return_labels=False)
    t4 = time.perf_counter()
    for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)): T[k] += v
kms, n = hip.profile_read()
print("c3 per sweep [ms]:", {k: round(v / 10 * 1e3, 3) for k, v in T.items()}, "kernels", round(kms / n, 3),
      "total", round(sum(T.values()) / 10 * 1e3, 3))
