"""Diagnostic: the label-statistics pass on the labels a C3 bench run actually draws (count profile of the components, device time with the
slot-table kernel and with the round-2 kernel).    python tools/c3_label_stats.py [rows]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[2] == "child":
    import numpy as np, torch
    import bench
    from mimo_amd.engine import HipEngine
    cfg = bench.CONFIGS["c3"]; desc, N, D, K, mode = cfg
    N = int(float(sys.argv[1]))
    X = bench.make_data(N, D, K, seed=1337, device="cuda:0"); torch.cuda.synchronize()
    hip = HipEngine(0); hip.upload(X)
    model = bench.build_model(cfg, hip)
    S = hip.label_stats(np.random.default_rng(4242).integers(0, K, size=N).astype(np.int32), K)
    rng = np.random.Generator(np.random.Philox(99))
    for it in range(12):
        _, S = model.gibbs_iteration(hip, S, it, label_rng='philox', seed=2024, param_rng=rng, stats=True, return_labels=False)
    n = np.sort(S.n)[::-1]
    print(f"slots={os.environ.get('MIMO_LABEL_STATS_SLOTS', '1')}: components with rows: {(n > 0).sum()}, top shares: "
          + " ".join(f"{v / N:.3f}" for v in n[:8]) + f", components above 1 %: {(n > 0.01 * N).sum()}")
    lab = hip.get_labels()
    for it in range(2): hip.label_stats(lab, K)
    hip.profile(True); hip.profile_read(reset=True)
    for it in range(5): hip.label_stats(lab, K)
    ms, k = hip.profile_read(reset=True)
    print(f"   label statistics of those labels: {ms / k * 1e3:.1f} us")
else:
    rows = sys.argv[1] if len(sys.argv) > 1 else "1e7"
    for slots in ("0", "1"):
        subprocess.run([sys.executable, __file__, rows, "child"], env=dict(os.environ, MIMO_LABEL_STATS_SLOTS=slots))
