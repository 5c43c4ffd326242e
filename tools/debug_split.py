import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd.engine import HipEngine
from oracle import mimo_oracle as O
from scipy.special import logsumexp
rng = np.random.default_rng(123)
N, D, K = 3001, 4, 37
centres = 400. * rng.standard_normal((K, D))
lab = rng.integers(0, K, N)
Z = centres[lab] + rng.standard_normal((N, D))
W = np.stack(K * [np.eye(D)]) * rng.uniform(0.5, 2., K)[:, None, None]
b = np.einsum('kde,ke->kd', W, centres)
c = -0.5 * np.einsum('kd,kd->k', centres, b) + rng.standard_normal(K)
c[5] = -np.inf
eng = HipEngine(0); eng.upload(Z)
for K2 in (37, 50, 64, 65, 70, 100):
    c3, b3, W3 = (np.concatenate([v, v[:K2 - K]]) for v in (c, b, W))
    with np.errstate(invalid='ignore'):
        L3 = O.canonical_eval(Z, c3, b3, W3)
    L3[5] = -np.inf
    lse3 = logsumexp(L3, axis=0); R3 = np.exp(L3 - lse3)
    with np.errstate(invalid='ignore'):
        srl3 = float(np.nansum(np.where(R3 > 0, R3 * L3, 0.)))
    _, sc = eng.estep(c3, b3, W3, entropy_split=True)
    _, scr = eng.estep(c3, b3, W3, keep_resp=True, keep_logp=True)
    Rg, Lg = eng.get_resp(K2), eng.get_logp(K2)
    with np.errstate(invalid='ignore'):
        srl_tab = float(np.nansum(np.where(Rg > 1e-200, Rg * Lg, 0.)))
    print(K2, "sc1-ref", sc[1] - srl3, "keep:", scr[1] - srl3, "tables:", srl_tab - srl3, "nan in L3:", np.isnan(L3).sum(), "sc0", sc[0] - lse3.sum())
    per_row = np.nansum(np.where(R3 > 0, R3 * L3, 0.), axis=0)
    per_row_g = np.nansum(np.where(Rg > 1e-200, Rg * Lg, 0.), axis=0)
    d = per_row_g - per_row
    print("   worst rows", np.argsort(-np.abs(d))[:4], d[np.argsort(-np.abs(d))[:4]], "labels", lab[np.argsort(-np.abs(d))[:4]])
