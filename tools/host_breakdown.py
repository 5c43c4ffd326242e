"""Diagnostic: where a VI step's wall time goes beyond the fused kernel (host conjugate update,
canonical form, parameter upload, D2H of the statistics).  python tools/host_breakdown.py c2|c4|c5 [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mimo_amd.engine import HipEngine
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg = bench.CONFIGS[name]; desc, N, D, K, mode = cfg
if len(sys.argv) > 2: N = int(float(sys.argv[2]))
X = bench.make_data(N, D, K, seed=1337, device="cuda:0", ilr=(mode == "ilr")); torch.cuda.synchronize()
hip = HipEngine(0); hip.upload(X)
model = bench.build_model(cfg, hip)
S = hip.label_stats(np.random.default_rng(1).integers(0, K, size=N).astype(np.int32), K)
T = {k: 0.0 for k in ("update", "canonical", "launch", "prior_terms", "wait")}
hip.profile(True)
for it in range(13):
    if it == 3:
        T = {k: 0.0 for k in T}; hip.profile_read(reset=True)
    t0 = time.perf_counter(); model._update_from_stats(S, sample=False)
    t1 = time.perf_counter(); th = model.canonical_expected()
    t2 = time.perf_counter(); hip.estep_async(*th)
    t3 = time.perf_counter(); pt = model._vlb_prior_terms()
    t4 = time.perf_counter(); S, sc = hip.estep_wait()
    t5 = time.perf_counter()
    for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): T[k] += v
kms, n = hip.profile_read()
print(name, "per step [ms]:", {k: round(v / 10 * 1e3, 3) for k, v in T.items()}, "kernel", round(kms / n, 3),
      "total", round(sum(T.values()) / 10 * 1e3, 3))
