"""Diagnostic: the same passes through two source trees of this repository on ONE box (e.g. the previous round's build under
tools/variants/r02tree against the working tree): device time of one softmax pass and one Gibbs sweep per shape.
    python tools/ab_trees.py <other tree> "D,K[,structure] ..." [N]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 4 and sys.argv[4] == "child":
    tree = sys.argv[1]
    sys.path.insert(0, tree)
    import numpy as np
    from mimo_amd.engine import HipEngine
    N = int(float(sys.argv[3])); eng = HipEngine(0); out = []
    for sh in sys.argv[2].split():
        parts = sh.split(","); D, K = int(parts[0]), int(parts[1]); st = parts[2] if len(parts) > 2 else "full"
        rng = np.random.default_rng(0)
        Z = rng.standard_normal((N, D)); A = rng.standard_normal((K, D, D))
        W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D); b = rng.standard_normal((K, D)); c = rng.standard_normal(K)
        if st == "diag": W = W * np.eye(D)
        if st == "linear": W = np.broadcast_to(W[:1], W.shape).copy()
        eng.set_structure(st); eng.upload(Z)
        for gibbs in (False, True):
            run = (lambda it: eng.gibbs_labels(c, b, W, seed=1, sweep=it, return_labels=False)) if gibbs else (lambda it: eng.estep(c, b, W))
            for it in range(3): run(it)
            eng.profile(True); eng.profile_read(reset=True)
            for it in range(8): run(it)
            ms, n = eng.profile_read(reset=True); eng.profile(False)
            out.append(f"{sh}{'g' if gibbs else 'v'} {ms / n * 1e3:.0f}")
    print(os.path.basename(tree.rstrip("/")) or tree, " | ".join(out), flush=True)
else:
    N = sys.argv[3] if len(sys.argv) > 3 else "2e6"
    for tree in (sys.argv[1], ROOT):
        subprocess.run([sys.executable, __file__, tree, sys.argv[2], N, "child"])
