"""Diagnostic / tuning: time every narrow_kernel instantiation (mimo_narrow.hip) forced to 1, 2, 3 and 4 waves per SIMD and write
the timings that tools/narrow_occupancy.py turns into mimo_amd/csrc/mimo_narrow_occ.h.
    on the GPU box:  for l in 0 1; do for w in 1 2 3 4; do bash tools/build_variant.sh narrow_w${w}_l$l "-DMIMO_NARROW_FORCE_WAVES=$w -DMIMO_NARROW_FORCE_LEAN=$l" mimo_narrow.hip; done; done
                     python tools/narrow_tune.py > gpurun_out/narrow_tune.jsonl
    then here:       python tools/narrow_occupancy.py gpurun_out/narrow_tune.jsonl"""
import glob, json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VS = (1, 2, 3, 4, 6, 8, 10, 12, 13, 14, 16, 18, 20, 22, 24, 25, 26, 28, 30, 32)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from mimo_amd.engine import HipEngine
    N = 1_500_000
    eng = HipEngine(0)
    w, lean = int(sys.argv[2]), int(sys.argv[3])
    for D in (1, 2, 3, 4):
        rng = np.random.default_rng(D)
        Z = rng.standard_normal((N, D)); eng.upload(Z)
        for V in VS:
            K = 4 * V
            A = rng.standard_normal((K, D, D)); W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D)
            b = rng.standard_normal((K, D)); c = rng.standard_normal(K)
            for mode in (0, 1):
                if eng.plan(K, gibbs=bool(mode))["kind"] != "narrow":
                    continue
                run = (lambda it: eng.gibbs_labels(c, b, W, seed=1, sweep=it, return_labels=False)) if mode else (lambda it: eng.estep(c, b, W))
                for it in range(2): run(it)
                eng.profile(True); eng.profile_read(reset=True)
                for it in range(6): run(it)
                kern = eng.profile_kernels(); eng.profile_read(reset=True); eng.profile(False)
                us = kern["narrow_kernel"]["ms"] / kern["narrow_kernel"]["launches"] * 1e3
                print(json.dumps({"waves": w, "lean": lean, "V": V, "NSF": D, "mode": mode, "us": round(us, 2)}), flush=True)
else:
    for lean in (0, 1):
        for w in (1, 2, 3, 4):
            lib = os.path.join(ROOT, "tools", "variants", f"narrow_w{w}_l{lean}.so")
            env = dict(os.environ, MIMO_HIP_LIB=lib, MIMO_NARROW_MIN_K="1")
            subprocess.run([sys.executable, __file__, "child", str(w), str(lean)], env=env, check=True)
