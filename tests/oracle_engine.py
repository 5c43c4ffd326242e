"""TEST DOUBLE of mimo_amd.engine.HipEngine backed by the CPU oracle (oracle/mimo_oracle.py).

Lives under tests/ and is injected through the `engine=` arguments of the host classes so that the
host-side logic (canonical forms, conjugate updates, drivers, sharded all-reduce) can be tested
without a GPU.  It is never importable from the product package.
"""
import numpy as np
from scipy.special import logsumexp

from oracle import mimo_oracle as O
from mimo_amd.engine import SuffStats


class OracleEngine:
    device = 0

    def __init__(self, device=0):
        self.N = self.D = 0
        self.Z = None
        self.row0 = 0
        self._K = None
        self._resp = self._logp = self._lse = self._labels = None

    def spawn(self):
        return OracleEngine()

    def global_rows(self, n_local):
        return int(n_local)

    def upload(self, Z):
        self.Z = np.ascontiguousarray(Z, dtype=float)
        self.N, self.D = self.Z.shape
        bad = np.isnan(self.Z).any(axis=1)          # rows with NaN: zeroed here, dropped from every statistic (mimo_nan_info)
        self._bad = np.flatnonzero(bad)
        self.n_bad = len(self._bad)
        if self.n_bad:
            self.Z = self.Z.copy()
            self.Z[bad] = 0.
        self._mask = (~bad).astype(float)

    def nan_rows(self):
        return self._bad

    def set_row_offset(self, row0):
        self.row0 = int(row0)

    def set_stream(self, s):
        pass

    def set_structure(self, structure):
        # the double always computes the full block (diagonal callers read its diagonal); under 'linear' it hands
        # back what HipEngine does for tied blocks: no per-component second moments, only their pooled sum
        self.structure = structure

    def _tied(self, W):
        W = np.asarray(W)
        return getattr(self, 'structure', 'full') == 'linear' and np.array_equal(W, np.broadcast_to(W[:1], W.shape))

    def _xx_total(self):
        return self.Z.T @ self.Z

    def _stats(self, R, pooled=False):
        R = np.asarray(R, float)
        n, sx, sxx = O.packed_stats(self.Z, R * self._mask if self.n_bad else R)
        S = SuffStats(n, sx, None, np.sum(sxx, axis=0)) if pooled else SuffStats(n, sx, sxx)
        if self.n_bad:
            S.n_rows = n + np.sum(R[:, self._bad], axis=1)
        return S

    def estep(self, c, b, W, stats=True, keep_resp=False, keep_logp=False, keep_lse=False, entropy_split=False,
              row_weights=None):
        L = O.canonical_eval(self.Z, np.asarray(c, float), np.asarray(b, float), np.asarray(W, float))
        lse = logsumexp(L, axis=0) if self.N else np.zeros(0)
        R = np.exp(L - lse)
        self._K = L.shape[0]
        if keep_resp:
            self._resp = R
        if keep_logp:
            self._logp = L
        if keep_lse:
            self._lse = lse
        srl = float(np.sum(R * L))
        sc = np.array([float(np.sum(lse)), srl, float(np.sum(lse)) - srl])
        Rw = R if row_weights is None else R * np.asarray(row_weights, float).reshape(1, -1)
        return (self._stats(Rw, self._tied(W) and not (keep_logp or keep_lse)) if stats else None), sc

    def gibbs_labels(self, c, b, W, seed=0, sweep=0, u=None, stats=True, return_labels=True, keep_logp=False):
        L = O.canonical_eval(self.Z, np.asarray(c, float), np.asarray(b, float), np.asarray(W, float))
        if u is None:
            u = O.philox_uniforms(seed, self.row0 + np.arange(self.N), sweep)
        labels = O.sample_discrete_from_log(L, np.asarray(u).reshape(-1))
        self._labels, self._K = labels, L.shape[0]
        if keep_logp:
            self._logp = L
        S = self._stats(O.one_hot(labels, L.shape[0]), self._tied(W) and not keep_logp) if stats else None
        return (labels if return_labels else None), S

    def weighted_stats(self, resp=None, K=None):
        return self._stats(self._resp if resp is None else np.asarray(resp, float), getattr(self, 'structure', 'full') == 'linear')

    def label_stats(self, labels, K):
        labels = self._labels if labels is None else np.asarray(labels).astype(int)
        if labels.size and (labels.min() < 0 or labels.max() >= K):
            raise ValueError("labels out of range")
        return self._stats(O.one_hot(labels, K), getattr(self, 'structure', 'full') == 'linear')

    def random_resp_stats(self, K, seed=0):
        rows = self.row0 + np.arange(self.N)
        V = np.stack([O.philox_uniforms(seed, rows, k) for k in range(K)]) + 1.1102230246251565e-16 if self.N else np.zeros((K, 0))
        self._resp, self._K = V / np.sum(V, axis=0), K
        return self._stats(self._resp, getattr(self, 'structure', 'full') == 'linear')

    def sample_from_log(self, logp=None, K=None, u=None, seed=0, sweep=0, return_lognorms=False):
        L = self._logp if logp is None else np.asarray(logp, float)
        if u is None:
            u = O.philox_uniforms(seed, self.row0 + np.arange(L.shape[1]), sweep)
        labels = O.sample_discrete_from_log(L, np.asarray(u).reshape(-1))
        return (labels, logsumexp(L, axis=0)) if return_lognorms else labels

    def table_entropy(self, table=None):
        t = self._resp if table is None else np.asarray(table, float)
        with np.errstate(invalid='ignore', divide='ignore'):
            return float(-np.nansum(t * np.log(t)))

    def predict(self, c, b, W, M, Q, Cc, affine=True, mode='average', y=None, P=None, ld=None, variance='full'):
        f = lambda a: None if a is None else np.asarray(a, float)
        mu, covar, nlpd = O.predict_canonical(self.Z, f(c), f(b), f(W), f(M), f(Q), f(Cc), affine, mode, f(y), f(P), f(ld))
        if variance == 'diagonal':
            var = np.diagonal(covar, axis1=1, axis2=2).copy()
            return mu, (var, np.sqrt(var)), nlpd
        return mu, covar, nlpd

    def get_resp(self, K=None):
        return self._resp

    def get_logp(self, K=None):
        return self._logp

    def get_lse(self):
        return self._lse

    def get_labels(self):
        return self._labels
