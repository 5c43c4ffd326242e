"""Diagnostic: per-phase cycle shares of the fused tile loop (build with `make -C mimo_amd/csrc stamps`)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mimo_amd._lib as L
L.LIB_PATH = os.environ.get("STAMPS_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmimo_hip_stamps.so")
from mimo_amd.engine import HipEngine
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 16
K = int(sys.argv[3]) if len(sys.argv) > 3 else 64
GIBBS = len(sys.argv) > 4 and sys.argv[4] == "gibbs"
rng = np.random.default_rng(0)
Z = rng.standard_normal((N, D)); A = rng.standard_normal((K, D, D))
W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D); b = rng.standard_normal((K, D)); c = rng.standard_normal(K)
eng = HipEngine(0); eng.upload(Z)
for it in range(3):
    if GIBBS: eng.gibbs_labels(c, b, W, seed=1, sweep=it, return_labels=False)
    else: eng.estep(c, b, W)
out = (C.c_double * 8)()
lib = L.load(); lib.mimo_debug_stamps.argtypes = [C.POINTER(C.c_double)]
assert lib.mimo_debug_stamps(out) == 0
names = ["wait B0 (top barrier)", "feature build", "wait B2", "E-step MFMA + Lt write", "wait B3", "normalise (softmax)",
         "wait B4", "stats MFMA + z staging"]
grid = lib.mimo_debug_stamps_grid()
tot = sum(out); ntile = (N + 31) // 32 / grid
print(f"cycles per wave per tile: total {tot/ntile:.0f}")
for n, v in zip(names, out): print(f"  {n:28s} {v/ntile:8.0f}  {100*v/tot:5.1f}%")
tr = (C.c_ulonglong * 128)()
if hasattr(lib, "mimo_debug_stamps_trace"):
    lib.mimo_debug_stamps_trace.argtypes = [C.POINTER(C.c_ulonglong)]
    if lib.mimo_debug_stamps_trace(tr) == 0 and any(tr):
        t0 = min(v for v in tr if v)
        print("wave 0 of workgroups 0 and grid/2: iteration: [tile start, build done + barrier, normalise done + barrier, tile end] cycles")
        for it in range(16):
            a = [tr[4 * it + i] - t0 for i in range(4)]; b = [tr[64 + 4 * it + i] - t0 for i in range(4)]
            print(f"  it {it + 8:2d}  A {a[0]:7d} {a[1]:7d} {a[2]:7d} {a[3]:7d}   B {b[0]:7d} {b[1]:7d} {b[2]:7d} {b[3]:7d}")
