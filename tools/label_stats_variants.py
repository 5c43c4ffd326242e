"""Diagnostic: the label-statistics kernels of every library variant under tools/variants/ (MIMO_HIP_LIB) next to the built library, one box:
device time of the statistics launch(es) of mimo_label_stats on a caller's label vector — uniform labels, and a skewed vector with a
geometric-like weight over `active` components (what a DP-GMM sweep at Kmax = 256 looks like after a few sweeps).
    python tools/label_stats_variants.py "D,K D,K ..." [N] [active]"""
import glob, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 4 and sys.argv[4] == "child":
    from mimo_amd.engine import HipEngine
    N, active = int(float(sys.argv[2])), int(sys.argv[3])
    eng = HipEngine(0)
    for sh in sys.argv[1].split():
        D, K = map(int, sh.split(","))
        rng = np.random.default_rng(0)
        Z = rng.standard_normal((N, D)); eng.upload(Z)
        out = []
        w = np.zeros(K); w[rng.permutation(K)[:min(active, K)]] = rng.random(min(active, K)) ** 2 + 0.02
        for name, lab in (("uniform", rng.integers(0, K, size=N)), (f"{min(active, K)} active, uneven", rng.choice(K, size=N, p=w / w.sum()))):
            lab = lab.astype(np.int32)
            for it in range(3): eng.label_stats(lab, K)
            eng.profile(True); eng.profile_read(reset=True)
            for it in range(10): eng.label_stats(lab, K)
            kern = eng.profile_kernels(); ms, n = eng.profile_read(reset=True); eng.profile(False)
            out.append(f"{name}: {ms / n * 1e3:7.1f} us = {N * (8 * D + 4) / (ms / n * 1e-3) / 1e12:5.2f} TB/s")
        print(f"{os.path.basename(os.environ.get('MIMO_HIP_LIB', 'built library')):22s} D={D:2d} K={K:3d} N={N}: " + " | ".join(out), flush=True)
else:
    N = sys.argv[2] if len(sys.argv) > 2 else "1e7"
    active = sys.argv[3] if len(sys.argv) > 3 else "150"
    for lib in [None] + sorted(glob.glob(os.path.join(ROOT, "tools", "variants", "*.so"))):
        env = dict(os.environ)
        if lib: env["MIMO_HIP_LIB"] = lib
        subprocess.run([sys.executable, __file__, sys.argv[1], N, active, "child"], env=env)
