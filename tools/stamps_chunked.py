"""Diagnostic: per-phase cycle shares of the chunked E-step (two-stage path), stamps build (`make -C mimo_amd/csrc stamps`)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mimo_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmimo_hip_stamps.so")
from mimo_amd.engine import HipEngine
N = int(float(sys.argv[1])); D = int(sys.argv[2]); K = int(sys.argv[3])
rng = np.random.default_rng(0)
Z = rng.standard_normal((N, D)); A = rng.standard_normal((K, D, D))
W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D); b = rng.standard_normal((K, D)); c = rng.standard_normal(K)
eng = HipEngine(0); eng.upload(Z)
for it in range(2): eng.estep(c, b, W)
lib = L.load(); lib.mimo_debug_stamps.argtypes = [C.POINTER(C.c_double)]
grid = lib.mimo_debug_stamps_grid()
ntile = (N + 31) // 32 / grid
for sel, names in ((2, ["chunk: MFMA steps + build hooks", "chunk: barrier wait", "softmax: local max (+ logp)", "barrier wait",
                        "softmax: exp + partial sums", "barrier wait", "softmax: totals, table write", "-"]),
                   (1, ["wait top barrier", "z staging (synchronous loads)", "chunks: wait barrier", "chunks: feature build",
                        "chunks: wait barrier 2", "chunks: MFMA", "Lt write + barrier", "normalise + table write"]),
                   (0, ["wait B0 (top barrier)", "feature build + weight tile load", "wait B2", "-", "-", "-", "-",
                        "stats MFMA + z staging"])):
    lib.mimo_debug_stamps_select(sel)
    out = (C.c_double * 8)()
    assert lib.mimo_debug_stamps(out) == 0
    tot = sum(out)
    if tot == 0: continue
    print(("wide E-step" if sel == 2 else "chunked E-step" if sel else "last statistics launch") + f": cycles per wave per tile: total {tot/ntile:.0f} (grid {grid})")
    for n, v in zip(names, out):
        if n != "-": print(f"  {n:34s} {v/ntile:8.0f}  {100*v/tot:5.1f}%")
tr = (C.c_ulonglong * 128)()
lib.mimo_debug_stamps_trace.argtypes = [C.POINTER(C.c_ulonglong)]
if lib.mimo_debug_stamps_trace(tr) == 0:
    t0 = min(v for v in tr if v)
    print("last statistics launch, wave 0 of workgroups 0 and grid/2 (one CU): iteration: [tile start, build done+barrier, (E-step end), MFMA end] cycles")
    for it in range(16):
        a = [tr[4 * it + i] - t0 for i in range(4)]; b = [tr[64 + 4 * it + i] - t0 for i in range(4)]
        print(f"  it {it + 8:2d}  A {a[0]:7d} {a[1]:7d} {a[3]:7d}   B {b[0]:7d} {b[1]:7d} {b[3]:7d}")
