"""Diagnostic (CPU, oracle-backed): could a WAVE-UNIFORM skip of dead exponentials speed up the C3 label kernel?  Runs the bench's own C3 workload
(bench.make_data / build_model, Gibbs sweeps from the seeded random start) on 32 000 rows with the oracle engine and counts, for the sweeps the bench
times, (i) the (row, component) pairs whose l - max < -707 (their exponential is the clamp value: dead) and (ii) the chunks of 8 / 16 / 32 components
that are dead for ALL 16 rows of a wave step — in the kernel's component layout and in contiguous layouts.  Result (profiles/r04_c3_exp_skip_potential.txt):
55 - 77 % of the exponentials are dead, 0.0 % of the chunks: rows of one step belong to different clusters.  VERDICT round 3, item 5."""
import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import bench
from oracle_engine import OracleEngine
from oracle import mimo_oracle as O
cfg = bench.CONFIGS["c3"]
desc, N, D, K, mode = cfg
N = 16 * 2000
X = bench.make_data(N, D, K, seed=1337, device="cpu").numpy()
eng = OracleEngine(); eng.upload(X)
model = bench.build_model(cfg, eng)
labels0 = np.random.default_rng(4242).integers(0, K, size=N).astype(np.int32)
S = eng.label_stats(labels0, K)
param_rng = np.random.Generator(np.random.Philox(99))
for it in range(26):
    _, S = model.gibbs_iteration(eng, S, it, label_rng='philox', seed=2024, param_rng=param_rng, stats=True, return_labels=False)
    if it in (0, 2, 5, 10, 15, 20, 25):
        c, b, W = model.likelihood.canonical()
        L = O.canonical_eval(X, c, b, W)          # (K, N)
        m = L.max(axis=0)
        dead = (L - m) < -707.0                    # (K, N)
        live_comp = (S.n > 0).sum()
        g = dead.reshape(K, N // 16, 16).all(axis=2)        # (K, groups): component dead for all 16 rows of the group
        # current layout: chunk c = components {64 q + 8 c + i}
        cur = np.stack([np.concatenate([g[64 * q + 8 * cc: 64 * q + 8 * cc + 8] for q in range(4)]).all(axis=0) for cc in range(8)]).mean()
        con32 = g.reshape(8, 32, -1).all(axis=1).mean()
        con16 = g.reshape(16, 16, -1).all(axis=1).mean()
        con8 = g.reshape(32, 8, -1).all(axis=1).mean()
        print(f"sweep {it}: occupied comps {live_comp}, dead (row,comp) frac {dead.mean():.3f}; wave-uniform dead chunks: current layout {cur:.3f}, contiguous-32 {con32:.3f}, contiguous-16 {con16:.3f}, contiguous-8 {con8:.3f}", flush=True)
