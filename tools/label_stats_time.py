"""Diagnostic: device time of the label-statistics pass (mimo_label_stats on a caller's label vector) for uniformly drawn labels and
for a skewed vector (all rows on `active` of the K components — what a DP-GMM sweep at Kmax = 256 looks like), with the slot-table
kernel and with the round-2 kernel (MIMO_LABEL_STATS_SLOTS=0 in a child process).
    python tools/label_stats_time.py "D,K D,K ..." [N] [active]"""
import os, subprocess, sys
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
        for name, lab in (("uniform", rng.integers(0, K, size=N)),
                          (f"{min(active, K)} active", rng.choice(rng.permutation(K)[:min(active, K)], size=N))):
            lab = lab.astype(np.int32)
            for it in range(2): eng.label_stats(lab, K)
            eng.profile(True); eng.profile_read(reset=True)
            for it in range(5): eng.label_stats(lab, K)
            ms, n = eng.profile_read(reset=True); eng.profile(False)
            out.append(f"{name}: {ms / n * 1e3:7.1f} us = {N * (8 * D + 4) / (ms / n * 1e-3) / 1e12:5.2f} TB/s")
        print(f"slots={os.environ.get('MIMO_LABEL_STATS_SLOTS', '1')} D={D:2d} K={K:3d} N={N}: " + " | ".join(out), flush=True)
else:
    N = sys.argv[2] if len(sys.argv) > 2 else "1e7"
    active = sys.argv[3] if len(sys.argv) > 3 else "32"
    for slots in ("0", "1"):
        subprocess.run([sys.executable, __file__, sys.argv[1], N, active, "child"], env=dict(os.environ, MIMO_LABEL_STATS_SLOTS=slots))
