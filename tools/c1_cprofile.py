"""Diagnostic: cProfile of the host side of the public mean-field loop at a small shape (which Python calls the 0.17 ms step is made of).
    python tools/c1_cprofile.py c1|sine [rows]"""
import cProfile, os, pstats, sys
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
for it in range(50): S, _ = model.meanfield_iteration(hip, S)
pr = cProfile.Profile(); pr.enable()
for it in range(2000): S, _ = model.meanfield_iteration(hip, S)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45)
