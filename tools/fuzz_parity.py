"""Fuzz: random (Dz, K, structure, N) through the softmax pass (plain, weighted, asynchronous) and the label pass (Philox and host
uniforms) against the oracle — statistics to 1e-10, labels exact (linear structure: up to last-bit ties), counts exact, a second
launch bit-identical.  Catches routing holes between the kernel families.
    python tools/fuzz_parity.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd.engine import HipEngine
from oracle import mimo_oracle as O
from scipy.special import logsumexp
def rel(a, b): return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def run(cases, seed, eng=None, max_rows=140003):
  """Returns the number of failing cases (prints them)."""
  rng = np.random.default_rng(seed)
  eng = eng or HipEngine(0)
  bad = 0
  for it in range(cases):
      D = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 16, 17, 20, 24, 27, 32]))
      K = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 17, 20, 24, 25, 32, 33, 40, 48, 50, 64, 65, 96, 100, 128, 129, 192, 200, 256]))
      st = str(rng.choice(["full", "full", "diag", "linear"]))
      N = min(int(rng.choice([1, 17, 4099, 33000, 140003])), max_rows)
      if N * K * D * D > 3e9: N = 4099                      # (the oracle's (K, N, D, D) temporaries)
      Z = rng.standard_normal((N, D)) * 1.5; A = rng.standard_normal((K, D, D))
      W = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D); mu = rng.standard_normal((K, D)) * 2
      if st == "diag": W = W * np.eye(D)
      if st == "linear": W = np.ascontiguousarray(np.broadcast_to(W[:1], W.shape))
      b = np.einsum('kde,ke->kd', W, mu); c = -0.5 * np.einsum('kd,kd->k', mu, b) + rng.standard_normal(K)
      tag = f"case {it}: Dz={D} K={K} {st} N={N}"
      try:
          eng.set_structure(st); eng.upload(Z)
          L = O.canonical_eval(Z, c, b, W); lse = logsumexp(L, axis=0); R = np.exp(L - lse)
          n, sx, sxx = O.packed_stats(Z, R)
          S, sc = eng.estep(c, b, W)
          S1, sc1 = eng.estep(c, b, W)
          w = rng.uniform(0., 2., size=N)
          Sw, scw = eng.estep(c, b, W, row_weights=w)
          wn, wsx, wsxx = O.packed_stats(Z, R * w[None, :])
          eng.estep_async(c, b, W); Sa, sca = eng.estep_wait()
          errs = [rel(S.n, n), rel(S.sx, sx), abs(sc[0] - lse.sum()) / max(1., abs(lse.sum())), rel(Sw.n, wn), rel(Sw.sx, wsx)]
          if st == "full": errs += [rel(S.sxx, sxx), rel(Sw.sxx, wsxx)]
          elif st == "diag": errs += [rel(S.sxx, sxx * np.eye(D)), rel(Sw.sxx, wsxx * np.eye(D))]
          else: errs += [rel(S.sxx_total, Z.T @ Z), rel(Sw.sxx_total, (Z * w[:, None]).T @ Z)]
          same = np.array_equal(S1.sx, S.sx) and sc1[0] == sc[0] and np.array_equal(Sa.sx, S.sx) and sca[0] == sc[0]
          u = rng.random(N)
          lab, G = eng.gibbs_labels(c, b, W, u=u)
          ref = O.sample_discrete_from_log(L, u)
          labp, Gp = eng.gibbs_labels(c, b, W, seed=11, sweep=it)
          refp = O.sample_discrete_from_log(L, O.philox_uniforms(11, np.arange(N), it))
          labq, Gq = eng.gibbs_labels(c, b, W, seed=11, sweep=it)
          tol = 1e-4 if st == "linear" else 0.0               # (the shared quadratic term only moves last-bit ties)
          lab_ok = np.mean(lab != ref) <= tol and np.mean(labp != refp) <= tol and np.array_equal(labq, labp)
          cnt_ok = np.array_equal(G.n, np.bincount(lab, minlength=K)) and np.array_equal(Gp.n, np.bincount(labp, minlength=K)) and np.array_equal(Gq.sx, Gp.sx)
          gsx = O.packed_stats(Z, O.one_hot(lab, K))[1]
          errs.append(rel(G.sx, gsx))
          ok = max(errs) < 1e-10 and same and lab_ok and cnt_ok
          plan = (eng.plan(K)["kind"], eng.plan(K, gibbs=True)["kind"])
          if not ok:
              bad += 1
              print("MISMATCH", tag, plan, "max err %.2e" % max(errs), "same", same, "labels", lab_ok, "counts", cnt_ok, flush=True)
          elif it % 20 == 0:
              print("ok", tag, plan, "max err %.1e" % max(errs), flush=True)
      except Exception as e:
          bad += 1
          print("EXCEPTION", tag, repr(e)[:300], flush=True)
      finally:
          eng.set_structure("full")
  print("cases:", cases, "bad:", bad)
  return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
