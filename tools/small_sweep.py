"""Diagnostic: kernel time (HIP events) and wall time of one pass for a list of small shapes, with the small-shape
VALU kernel and (MIMO_SMALL=0 in a child process) through the MFMA tile kernels.   python tools/small_sweep.py [N]"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(1, 4), (1, 6), (1, 16), (1, 32), (2, 4), (2, 6), (2, 8), (2, 16), (2, 25), (2, 32), (3, 4), (3, 8), (3, 16),
          (3, 32), (4, 4), (4, 8), (4, 16), (4, 32)]


def child(N):
    from mimo_amd.engine import HipEngine
    eng = HipEngine(0)
    for D, K in SHAPES:
        rng = np.random.default_rng(0)
        Z = rng.standard_normal((N, D)); A = rng.standard_normal((K, D, D))
        W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D); b = rng.standard_normal((K, D)); c = rng.standard_normal(K)
        eng.upload(Z)
        out = []
        for gibbs in (False, True):
            run = (lambda it: eng.gibbs_labels(c, b, W, seed=1, sweep=it, return_labels=False)) if gibbs else (lambda it: eng.estep(c, b, W))
            for it in range(3): run(it)
            eng.profile(True); eng.profile_read(reset=True)
            ts = []
            for it in range(10):
                t0 = time.perf_counter(); run(it); ts.append(time.perf_counter() - t0)
            ms, n = eng.profile_read(reset=True); eng.profile(False)
            out.append((ms / n, float(np.median(ts)) * 1e3))
        print(f"D={D} K={K:2d} {eng.plan(K)['kind']:6s} vi kernel {out[0][0]:.3f} wall {out[0][1]:.3f} ms | gibbs kernel {out[1][0]:.3f} wall {out[1][1]:.3f} ms"
              f" | vi {8 * N * D / out[0][0] / 1e6:.0f} GB/s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "child":
        child(int(float(sys.argv[1])))
    else:
        N = sys.argv[1] if len(sys.argv) > 1 else "1e7"
        for small in ("1", "0"):
            print(f"--- MIMO_SMALL={small}", flush=True)
            subprocess.run([sys.executable, __file__, N, "child"], env=dict(os.environ, MIMO_SMALL=small), check=True)
