"""Diagnostic: label pass of 129 .. 256 components at Dz = 3, 4 on the narrow kernels (MIMO_NARROW_BIG_LABELS_WIDE=1) against the
row-owner kernels that serve them.   python tools/narrow_big_labels_time.py [N]"""
import os, subprocess, sys
N = sys.argv[1] if len(sys.argv) > 1 else "2e6"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for D in (3, 4):
    for K in (144, 160, 192, 224, 256):
        outs = []
        for env in ({"MIMO_NARROW_BIG_LABELS_WIDE": "1"}, {}):
            e = dict(os.environ); e.update(env)
            r = subprocess.run([sys.executable, os.path.join(root, "tools", "quick_time.py"), N, str(D), str(K), "gibbs"], capture_output=True, text=True, env=e)
            line = [l for l in r.stdout.splitlines() if l.startswith("N=")]
            outs.append(line[0].split(":")[1].strip().split()[0] if line else "ERR " + r.stderr[-200:])
        print(f"Dz={D} K={K:3d} gibbs: narrow {outs[0]} ms | row-owner kernels {outs[1]} ms", flush=True)
