"""Diagnostic: 129 .. 256 components at Dz <= 2 on the narrow kernels (mimo_narrow_big.hip) against the tile kernels they replace
(MIMO_NARROW_MAX_K=128 in a child process).   python tools/narrow_big_time.py [N]"""
import os, subprocess, sys
N = sys.argv[1] if len(sys.argv) > 1 else "2e6"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for D in (1, 2):
    for K in (144, 160, 192, 224, 256):
        for mode in ("vi", "gibbs"):
            outs = []
            for env in ({}, {"MIMO_NARROW_MAX_K": "128"}):
                e = dict(os.environ); e.update(env)
                r = subprocess.run([sys.executable, os.path.join(root, "tools", "quick_time.py"), N, str(D), str(K)] + (["gibbs"] if mode == "gibbs" else []),
                                   capture_output=True, text=True, env=e)
                line = [l for l in r.stdout.splitlines() if l.startswith("N=")]
                outs.append(line[0].split(":")[1].strip().split()[0] if line else "ERR " + r.stderr[-200:])
            print(f"Dz={D} K={K:3d} {mode:5s}: narrow {outs[0]} ms | tile kernels {outs[1]} ms", flush=True)
