"""Diagnostic: kernel time of a few shapes for every library variant under tools/variants/ (MIMO_HIP_LIB).
    python tools/variant_time.py "D,K D,K ..." [N]"""
import glob, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 3 and sys.argv[3] == "child":
    from mimo_amd.engine import HipEngine
    N = int(float(sys.argv[2])); eng = HipEngine(0); line = []
    for sh in sys.argv[1].split():
        D, K = map(int, sh.split(","))
        rng = np.random.default_rng(0)
        Z = rng.standard_normal((N, D)); A = rng.standard_normal((K, D, D))
        W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D); b = rng.standard_normal((K, D)); c = rng.standard_normal(K)
        eng.upload(Z)
        for gibbs in (False, True):
            run = (lambda it: eng.gibbs_labels(c, b, W, seed=1, sweep=it, return_labels=False)) if gibbs else (lambda it: eng.estep(c, b, W))
            for it in range(3): run(it)
            eng.profile(True); eng.profile_read(reset=True)
            for it in range(10): run(it)
            ms, n = eng.profile_read(reset=True); eng.profile(False)
            line.append(f"{sh}{'g' if gibbs else 'v'} {ms / n * 1e3:.1f}us")
    print(os.path.basename(os.environ.get("MIMO_HIP_LIB", "default")), " | ".join(line), flush=True)
else:
    N = sys.argv[2] if len(sys.argv) > 2 else "1e7"
    for lib in [None] + sorted(glob.glob(os.path.join(ROOT, "tools", "variants", "*.so"))):
        env = dict(os.environ)
        if lib: env["MIMO_HIP_LIB"] = lib
        subprocess.run([sys.executable, __file__, sys.argv[1], N, "child"], env=env)
