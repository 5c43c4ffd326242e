"""Diagnostic: device time of predict_kernel (mimo_predict: posterior-predictive mixture moments, one thread per row) and its
fraction of the float64 VALU rate, with device-resident inputs and outputs (mimo_predict_flags) — and the wall time of the
host-array form next to it (PCIe both ways).
    python tools/predict_time.py [N]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd.engine import HipEngine
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000
eng = HipEngine(0)
rng = np.random.default_rng(0)
for dx, dy, K in ((1, 1, 50), (1, 1, 100), (2, 1, 64), (8, 4, 64), (3, 2, 16)):
    dc = dx + 1
    Z = rng.standard_normal((N, dx)) * 2
    A = rng.standard_normal((K, dx, dx)); W = A @ A.transpose(0, 2, 1) / dx + 0.3 * np.eye(dx)
    b = rng.standard_normal((K, dx)); c = rng.standard_normal(K)
    M = rng.standard_normal((K, dy, dc))
    A = rng.standard_normal((K, dc, dc)); Q = A @ A.transpose(0, 2, 1) / dc + 0.1 * np.eye(dc)
    B = rng.standard_normal((K, dy, dy)); Cc = B @ B.transpose(0, 2, 1) / dy + 0.2 * np.eye(dy)
    Zd = torch.from_numpy(Z).cuda(); eng.upload(Zd)
    mu = torch.empty((N, dy), dtype=torch.float64, device="cuda"); cov = torch.empty((N, dy, dy), dtype=torch.float64, device="cuda")
    for it in range(2): eng.predict_device(c, b, W, M, Q, Cc, mu.data_ptr(), cov.data_ptr())
    torch.cuda.synchronize(); eng.estep(c, b, W, stats=False)
    eng.profile(True); eng.profile_read(reset=True)
    for it in range(5): eng.predict_device(c, b, W, M, Q, Cc, mu.data_ptr(), cov.data_ptr())
    kern = eng.profile_kernels(); eng.profile_read(reset=True); eng.profile(False)
    ms = kern["predict_kernel"]["ms"] / kern["predict_kernel"]["launches"]
    # multiply-adds per (row, component): gate quadratic form dx^2 + dx, x~'Qx~ dc^2, mean dy dc, covariance dy^2 (+ the softmax)
    flops = 2.0 * N * K * (dx * dx + dx + dc * dc + dy * dc + dy * dy)
    t0 = time.perf_counter(); eng.upload(Z); eng.predict(c, b, W, M, Q, Cc); wall = time.perf_counter() - t0
    print(f"N={N} dx={dx} dy={dy} K={K}: predict_kernel {ms:7.3f} ms = {N * K / ms / 1e6:8.1f} G row-components/s, "
          f"{flops / ms / 1e9:6.2f} TFLOP/s = {flops / ms / 1e9 / 78.6:5.3f} of the f64 rate, "
          f"{N * 8 * (dx + dy + dy * dy) / ms / 1e9:5.2f} TB/s of rows in / moments out; host-array form {wall * 1e3:7.1f} ms wall", flush=True)
