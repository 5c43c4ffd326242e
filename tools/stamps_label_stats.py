"""Diagnostic: per-phase cycles of the label-statistics kernel of a Gibbs sweep (stamps build: make -C mimo_amd/csrc stamps).
    python tools/stamps_label_stats.py N D K        (MIMO_LABEL_STATS_BITS=0: the round-2 kernel)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mimo_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmimo_hip_stamps.so")
from mimo_amd.engine import HipEngine
N = int(float(sys.argv[1])); D = int(sys.argv[2]); K = int(sys.argv[3])
rng = np.random.default_rng(0)
Z = rng.standard_normal((N, D))
labels = rng.integers(0, K, size=N).astype(np.int32)
eng = HipEngine(0); eng.upload(Z)
for it in range(3): eng.label_stats(labels, K)
out = (C.c_double * 8)()
lib = L.load(); lib.mimo_debug_stamps.argtypes = [C.POINTER(C.c_double)]
assert lib.mimo_debug_stamps(out) == 0
names = ["wait: top barrier", "z tile + bitmap zero -> LDS, next loads", "wait: barrier", "atomic OR + barrier",
         "popcount + prefix scan (+ barriers)", "place rows in the list + barrier", "accumulate", "-"]
grid = lib.mimo_debug_stamps_grid()
tot = sum(out)
print(f"N={N} D={D} K={K} bits={os.environ.get('MIMO_LABEL_STATS_BITS', '1')}: cycles per wave, whole kernel: {tot:.0f} (grid {grid})")
for n, v in zip(names, out):
    if n != "-": print(f"  {n:44s} {v:10.0f}  {100 * v / tot:5.1f}%")
