"""Diagnostic: per-phase cycle shares of the statistics-only launches (two-stage path), stamps build."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mimo_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmimo_hip_stamps.so")
from mimo_amd.engine import HipEngine
N = int(float(sys.argv[1])); D = int(sys.argv[2]); K = int(sys.argv[3])
rng = np.random.default_rng(0)
Z = rng.standard_normal((N, D))
R = rng.random((K, N)); R /= R.sum(0)
eng = HipEngine(0); eng.upload(Z)
for it in range(2): eng.weighted_stats(R)
out = (C.c_double * 8)()
lib = L.load(); lib.mimo_debug_stamps.argtypes = [C.POINTER(C.c_double)]
assert lib.mimo_debug_stamps(out) == 0
names = ["wait B0 (top barrier)", "feature build + weight tile load", "wait B2", "-", "-", "-", "-", "stats MFMA + z staging"]
grid = lib.mimo_debug_stamps_grid()
tot = sum(out); ntile = (N + 31) // 32 / grid
print(f"last statistics launch: cycles per wave per tile: total {tot/ntile:.0f} (grid {grid})")
for n, v in zip(names, out):
    if n != "-": print(f"  {n:34s} {v/ntile:8.0f}  {100*v/tot:5.1f}%")
