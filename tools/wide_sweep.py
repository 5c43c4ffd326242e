"""Diagnostic: device time of one VI pass and one Gibbs sweep for FEW components over many features (K <= 32, Dz = 5 .. 16), with the
table-driven narrow kernels (mimo_narrow.hip) and without them (MIMO_NARROW_WIDE=0 in a child process: row-owner / tile kernels).
    python tools/wide_sweep.py [N]"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
Ds = tuple(int(x) for x in os.environ.get("WIDE_SWEEP_DS", "5,6,8,10,12,14,16").split(","))
Ks = tuple(int(x) for x in os.environ.get("WIDE_SWEEP_KS", "1,2,4,8,12,16,24,32").split(","))
if len(sys.argv) > 2 and sys.argv[2] == "child":
    from mimo_amd.engine import HipEngine
    N = int(float(sys.argv[1]))
    eng = HipEngine(0)
    rng = np.random.default_rng(0)
    for mode in os.environ.get("WIDE_SWEEP_MODES", "vi,gibbs").split(","):
        print(f"{mode} narrow_wide={os.environ.get('MIMO_NARROW_WIDE', '1')} max_k={os.environ.get('MIMO_NARROW_WIDE_MAX_K', 'default')}: us per pass / route (N = {N}); rows Dz, columns K = {Ks}")
        for D in Ds:
            Z = rng.standard_normal((N, D)); eng.upload(Z)
            line = []
            for K in Ks:
                A = rng.standard_normal((K, D, D)); W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D)
                b = rng.standard_normal((K, D)); c = rng.standard_normal(K)
                run = (lambda it: eng.gibbs_labels(c, b, W, seed=1, sweep=it, return_labels=False)) if mode == "gibbs" else (lambda it: eng.estep(c, b, W))
                for it in range(2): run(it)
                eng.profile(True); eng.profile_read(reset=True)
                for it in range(4): run(it)
                ms, n = eng.profile_read(reset=True); eng.profile(False)
                line.append(f"{ms / n * 1e3:7.1f}/{eng.plan(K, gibbs=(mode == 'gibbs')).get('kind', '?')[:4]}")
            print(f"  Dz={D:2d} " + " ".join(line), flush=True)
else:
    N = sys.argv[1] if len(sys.argv) > 1 else "2e6"
    for env in ({"MIMO_NARROW_WIDE": "0"}, {"MIMO_NARROW_WIDE": "1"}):
        subprocess.run([sys.executable, __file__, N, "child"], env=dict(os.environ, **env))
