"""First-light check of libmimo_hip.so against a direct NumPy evaluation of the canonical form
(independent of oracle/): run on the GPU box with `python tools/first_light.py`."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mimo_amd.engine import HipEngine, philox_uniforms


def direct(Z, c, b, W):
    L = c[:, None] + b @ Z.T - 0.5 * np.einsum('nd,kde,ne->kn', Z, W, Z)
    m = L.max(0)
    lse = m + np.log(np.exp(L - m).sum(0))
    R = np.exp(L - lse)
    return L, lse, R


def stats(Z, R):
    return R.sum(1), R @ Z, np.einsum('kn,nd,ne->kde', R, Z, Z)


def rel(a, b):
    if a.size == 0 and b.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


def run(N, D, K, rng, eng):
    Z = rng.standard_normal((N, D)) * 2.0 + rng.standard_normal(D)
    A = rng.standard_normal((K, D, D))
    W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D)
    mu = rng.standard_normal((K, D)) * 2
    b = np.einsum('kde,ke->kd', W, mu)
    c = -0.5 * np.einsum('kd,kd->k', mu, b) + rng.standard_normal(K) * 0.1
    eng.upload(Z)
    L, lse, R = direct(Z, c, b, W)
    S, sc = eng.estep(c, b, W, keep_resp=True, keep_logp=True, keep_lse=True)
    n, sx, sxx = stats(Z, R)
    errs = dict(logp=rel(eng.get_logp(), L), lse=rel(eng.get_lse(), lse), resp=rel(eng.get_resp(), R),
                n=rel(S.n, n), sx=rel(S.sx, sx), sxx=rel(S.sxx, sxx),
                s0=abs(sc[0] - lse.sum()) / max(1e-300, abs(lse.sum())), s1=abs(sc[1] - (R * L).sum()) / max(1e-300, abs((R * L).sum())))
    # weighted stats with arbitrary weights
    Wt = rng.random((K, N)); Wt /= Wt.sum(0)
    S2 = eng.weighted_stats(Wt)
    n2, sx2, sxx2 = stats(Z, Wt)
    errs.update(w_n=rel(S2.n, n2), w_sx=rel(S2.sx, sx2), w_sxx=rel(S2.sxx, sxx2))
    # gibbs with host uniforms
    u = rng.random(N)
    P = np.exp(L - lse); cum = np.cumsum(P, axis=0)
    ref_lab = np.sum(u * cum[-1] > cum, axis=0, dtype=np.int32)
    lab, S3 = eng.gibbs_labels(c, b, W, u=u)
    oh = np.zeros((K, N)); oh[ref_lab, np.arange(N)] = 1
    n3, sx3, sxx3 = stats(Z, oh)
    errs.update(label_flips=int((lab != ref_lab).sum()), g_n=rel(S3.n, n3), g_sxx=rel(S3.sxx, sxx3))
    # philox mode
    up = philox_uniforms(7, np.arange(N), 3)
    lab_p, _ = eng.gibbs_labels(c, b, W, seed=7, sweep=3, stats=False)
    ref_p = np.sum(up * cum[-1] > cum, axis=0, dtype=np.int32)
    errs.update(philox_flips=int((lab_p != ref_p).sum()))
    S4 = eng.label_stats(ref_lab, K)
    errs.update(l_sxx=rel(S4.sxx, sxx3))
    bad = {k: v for k, v in errs.items() if (v > 1e-9 if 'flips' not in k else v > 0)}
    print(f"N={N} D={D} K={K}: " + ("OK" if not bad else f"FAIL {bad}"), {k: (f"{v:.1e}" if isinstance(v, float) else v) for k, v in errs.items()})
    return not bad


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    eng = HipEngine(0)
    ok = True
    for (N, D, K) in [(257, 2, 4), (1000, 3, 5), (4099, 8, 32), (4099, 8, 256), (4099, 12, 64), (4099, 16, 64),
                      (31, 16, 16), (33, 5, 70), (100000, 16, 64), (0, 4, 3)]:
        ok &= run(N, D, K, rng, eng)
    # timing at C2 size if requested
    if len(sys.argv) > 1:
        N = int(float(sys.argv[1])); D, K = 16, 64
        Z = rng.standard_normal((N, D))
        A = rng.standard_normal((K, D, D)); W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D)
        b = rng.standard_normal((K, D)); c = rng.standard_normal(K)
        eng.upload(Z); eng.profile(True)
        for _ in range(2): eng.estep(c, b, W)
        eng.profile_read()
        t0 = time.time()
        for _ in range(5): eng.estep(c, b, W)
        dt = (time.time() - t0) / 5
        ms, n = eng.profile_read()
        print(f"C2-shape N={N}: wall {dt*1e3:.2f} ms/sweep, kernel {ms/n:.2f} ms => {N*K/(ms/n*1e-3):.3e} evals/s (kernel), {N*K/dt:.3e} (wall)")
    sys.exit(0 if ok else 1)
