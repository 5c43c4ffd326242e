"""(Bayesian) mixtures of linear-Gaussian experts ("infinite local regression").

Same method surface as the reference's mimo/mixtures/ilr.py (inference part, :21-323).  The joint
row z = [x, y] is the datum on the engine: the input density over x (`basis`), the expert density of
y | x (`models`) and the gating add up to ONE quadratic form in z per component, and one fused pass
returns the statistics of all three conjugate blocks (they are blocks of sum_n r z~ z~').
"""
import numpy as np
import numpy.random as npr
from tqdm import tqdm

from mimo_amd import engine as _engine
from mimo_amd.utils.abstraction import Statistics as Stats
from mimo_amd.utils.data import batches
from scipy.special import logsumexp

from mimo_amd.distributions.lingauss import split_joint_stats, joint_rows, canonical_rows, nan_row_sets
from mimo_amd.mixtures.gmm import canonical_inner, random_start, LazyTable
from mimo_amd.mixtures import _svi


class Standardizer:
    """Column standardisation with sklearn.preprocessing.StandardScaler semantics (population std,
    zero-variance columns left unscaled) — what ilr.py:108-109,124-127 uses."""

    def fit(self, a):
        a = np.asarray(a, dtype=float)
        self.mean_ = a.mean(axis=0)
        self.var_ = a.var(axis=0)
        self.scale_ = a.std(axis=0)
        self.scale_[self.scale_ == 0.] = 1.
        return self

    def transform(self, a):
        return (np.asarray(a, dtype=float) - self.mean_) / self.scale_

    def inverse_transform(self, a):
        return np.asarray(a, dtype=float) * self.scale_ + self.mean_


def embed_joint(basis_cbw, models_cbw, gating_log, dx):
    """Sum of the three per-component terms as one (c, b, W) over z = [x, y]."""
    c1, b1, W1 = basis_cbw
    c2, b2, W2 = models_cbw
    W = W2.copy()
    b = b2.copy()
    W[:, :dx, :dx] += W1
    b[:, :dx] += b1
    return c1 + c2 + gating_log, b, W


class MixtureOfLinearGaussians:
    """reference: mimo/mixtures/ilr.py:21-84"""

    def __init__(self, size, input_dim, output_dim, gating, basis, models, engine=None):
        self.size = size
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.gating = gating
        self.basis = basis
        self.models = models
        self._engine = engine

    @property
    def engine(self):
        return self._engine if self._engine is not None else self.models.engine

    def canonical(self):
        with np.errstate(divide='ignore'):
            return embed_joint(self.basis.canonical(), self.models.canonical(),
                               np.log(self.gating.probs), self.input_dim)

    def _bind(self, x, y):
        x = np.asarray(x, dtype=float).reshape(-1, self.input_dim)
        y = np.asarray(y, dtype=float).reshape(-1, self.output_dim)
        return _engine.bind(self.engine, joint_rows(x, y))

    def nan_rows_table(self, eng, x, y):
        """None, or (rows, (K, n) table) for the rows that hold a NaN — the reference's element-wise rules instead of the engine's
        whole-row value (z = 0).  Inside log_complete_likelihood (ilr.py:71-75 of the reference) the input density runs first:
        a row with a NaN in x gets its normaliser only (gaussian.py:512-520) — and nan_to_num's x IN PLACE, so the experts'
        density that runs next sees no NaN in x, zeroes nothing, and evaluates every row on the nan_to_num'ed (x, y)."""
        if not getattr(eng, 'n_bad', 0):
            return None
        x = np.asarray(x, dtype=float).reshape(-1, self.input_dim)
        y = np.asarray(y, dtype=float).reshape(-1, self.output_dim)
        bad, bx, by = nan_row_sets(x, y)
        cb, bb, Wb = self.basis.canonical()
        basis = canonical_rows(cb, bb, Wb, np.nan_to_num(x[bad]))
        basis[:, bx] = cb[:, None]
        models = self.models.nan_rows_loglik(x[bad], y[bad], bx, by, zero_when_both=False)
        with np.errstate(divide='ignore'):
            return bad, basis + models + np.log(self.gating.probs)[:, None]

    def log_complete_likelihood(self, x, y):
        eng = self._bind(x, y)
        eng.estep(*self.canonical(), stats=False, keep_logp=True)
        L = eng.get_logp(self.size)
        fix = self.nan_rows_table(eng, x, y)
        if fix is not None:
            L[:, fix[0]] = fix[1]
        return L

    def log_likelihood(self, x, y):
        eng = self._bind(x, y)
        eng.estep(*self.canonical(), stats=False, keep_lse=True)
        lse = eng.get_lse()
        fix = self.nan_rows_table(eng, x, y)
        if fix is not None:
            lse[fix[0]] = logsumexp(fix[1], axis=0)
        return lse

    def responsibilities(self, x, y):
        eng = self._bind(x, y)
        eng.estep(*self.canonical(), stats=False, keep_resp=True)
        R = eng.get_resp(self.size)
        fix = self.nan_rows_table(eng, x, y)
        if fix is not None:
            R[:, fix[0]] = np.exp(fix[1] - logsumexp(fix[1], axis=0))
        return R

    def rvs(self, size=1):
        z = self.gating.rvs(size)
        counts = np.bincount(z, minlength=self.size)
        x = np.empty((size, self.input_dim))
        y = np.empty((size, self.output_dim))
        bci, mci = self.basis.lmbdas_chol_inv, self.models.lmbdas_chol_inv
        for idx, count in enumerate(counts):
            shape = self.input_dim if count == 1 else (count, self.input_dim)
            xk = np.reshape(self.basis.mus[idx] + npr.normal(size=shape).dot(bci[idx].T), (-1, self.input_dim))
            mu = self.models.predict(xk)[idx]
            x[z == idx, ...] = xk
            y[z == idx, ...] = mu + npr.normal(size=(xk.shape[0], self.output_dim)).dot(mci[idx].T)
        perm = npr.permutation(size)
        return x[perm], y[perm], z[perm]


class BayesianMixtureOfLinearGaussians:
    """reference: mimo/mixtures/ilr.py:87-430."""

    def __init__(self, size, input_dim, output_dim, gating, basis, models, scale=False, engine=None):
        self.size = size
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.gating = gating
        self.basis = basis
        self.models = models
        self.likelihood = MixtureOfLinearGaussians(size, input_dim, output_dim, gating=self.gating.likelihood,
                                                   basis=self.basis.likelihood, models=self.models.likelihood,
                                                   engine=engine)
        self.scale = scale
        self.input_transform = Standardizer()
        self.output_transform = Standardizer()
        self._engine = engine
        self._batch_engine = None
        self.labels_ = None

    @property
    def engine(self):
        return self._engine if self._engine is not None else self.models.likelihood.engine

    @property
    def affine(self):
        return self.models.likelihood.affine

    def init_transform(self, x, y):
        self.scale = True
        self.input_transform.fit(x)
        self.output_transform.fit(y)

    def _scaled(self, x, y):
        x = np.asarray(x, dtype=float).reshape(-1, self.input_dim)
        y = np.asarray(y, dtype=float).reshape(-1, self.output_dim)
        if self.scale:
            key = (x.__array_interface__['data'][0], y.__array_interface__['data'][0], x.shape)
            hit = getattr(self, "_scaled_cache", None)
            if hit is None or hit[0] != key:
                hit = (key, np.ascontiguousarray(self.input_transform.transform(x)),
                       np.ascontiguousarray(self.output_transform.transform(y)), x, y)
                self._scaled_cache = hit
            return hit[1], hit[2]
        return x, y

    def _bind(self, xx, yy):
        return _engine.bind(self.engine, joint_rows(xx, yy))

    def used_labels(self, x, y):
        labels = np.argmax(self.expected_responsibilities(*self._scaled(x, y)), axis=0)
        return np.where(np.bincount(labels, minlength=self.size) > 0)[0]

    # ---- statistics blocks -----------------------------------------------------------------------
    def _block_stats(self, S):
        (xk, xxTk), (yxT, xxT, yyT) = split_joint_stats(S, self.input_dim, self.affine)
        return Stats([xk, S.n, xxTk, S.n]), Stats([yxT, xxT, yyT, S.n])

    # ---- canonical forms --------------------------------------------------------------------------
    def canonical_expected(self):
        """ilr.py:178-189: basis + models + gating, mean-field form."""
        return embed_joint(self.basis.canonical_expected(), self.models.canonical_expected(),
                           self.gating.expected_log_gating(), self.input_dim)

    # ---- Gibbs sampling --------------------------------------------------------------------------
    def resample(self, x, y, init_labels='prior', maxiter=1, progress_bar=True, process_id=0,
                 label_rng='host', seed=0, param_rng=None):
        """ilr.py:134-159 — sweep order basis -> models -> gating -> labels."""
        xx, yy = self._scaled(x, y)
        eng = self._bind(xx, yy)
        N = eng.N
        if init_labels == 'random':
            z = npr.choice(self.size, size=(N))
        elif init_labels == 'posterior':
            z = self._draw_labels(eng, label_rng, seed, 0, stats=False)[0]
        elif init_labels == 'prior':
            z = self.gating.likelihood.rvs(N)
        else:
            raise ValueError(init_labels)
        S = eng.label_stats(z, self.size)
        with tqdm(total=maxiter, desc=f'Init #{process_id + 1}', position=process_id,
                  disable=not progress_bar) as pbar:
            for it in range(maxiter):
                bstats, mstats = self._block_stats(S)
                self.basis.resample(None, stats=bstats, rng=param_rng)
                self.models.resample(None, None, stats=mstats, rng=param_rng)
                self.gating.resample(None, counts=S.gating_counts)
                last = it == maxiter - 1
                z, S = self._draw_labels(eng, label_rng, seed, it + 1, stats=not last, return_labels=last)
                pbar.update(1)
        self.labels_ = z

    resample_model = resample

    def _draw_labels(self, eng, label_rng, seed, sweep, stats=True, return_labels=True):
        c, b, W = self.likelihood.canonical()
        if hasattr(eng, 'check_replicated_once'):      # sharded: every rank drew these blocks from ITS host generator
            eng.check_replicated_once(c, b, W, what="basis / model / gating parameters drawn for the label pass")
        if label_rng == 'host':
            return eng.gibbs_labels(c, b, W, u=npr.random(size=(1, eng.N)), stats=stats,
                                    return_labels=return_labels)
        if label_rng == 'philox':
            return eng.gibbs_labels(c, b, W, seed=seed, sweep=sweep, stats=stats, return_labels=return_labels)
        raise ValueError(label_rng)

    def resample_labels(self, x, y, lazy=True):
        """ilr.py:161-164 -> (log_prob, labels); the table is lazy by default (see gmm.py, resample_labels)."""
        xx, yy = self._as2d(x, y)
        eng = self._bind(xx, yy)
        c, b, W = self.likelihood.canonical()
        u = npr.random(size=(1, eng.N))
        fix = self.likelihood.nan_rows_table(eng, xx, yy)      # rows with a NaN: the reference's element-wise rules (few rows, host)

        def redraw(labels):
            if fix is not None:
                rows, Lb = fix
                cum = np.cumsum(np.exp(Lb - logsumexp(Lb, axis=0)), axis=0)          # mimo/utils/stats.py:8-21 on those columns
                labels[rows] = np.sum(u[0, rows] * cum[-1] > cum, axis=0).astype(labels.dtype)
            return labels
        if not lazy:
            labels, _ = eng.gibbs_labels(c, b, W, u=u, stats=False, keep_logp=True)
            L = eng.get_logp(self.size)
            if fix is not None:
                L[:, fix[0]] = fix[1]
            return L, redraw(labels)
        labels, _ = eng.gibbs_labels(c, b, W, u=u, stats=False)
        c, b, W = np.array(c), np.array(b), np.array(W)

        def table():
            e = self._bind(xx, yy)
            e.estep(c, b, W, stats=False, keep_logp=True)
            L = e.get_logp(len(c))
            if fix is not None:
                L[:, fix[0]] = fix[1]
            return L
        return LazyTable(table, (len(c), eng.N)), redraw(labels)

    def _as2d(self, x, y):
        return (np.asarray(x, dtype=float).reshape(-1, self.input_dim),
                np.asarray(y, dtype=float).reshape(-1, self.output_dim))

    def resample_gating(self, z):
        self.gating.resample(np.asarray(z).astype(int))

    def resample_basis(self, x, z):
        """ilr.py:169-171 without the dense one_hot table."""
        S = self.basis.likelihood._bind(x).label_stats(z, self.size)
        self.basis.resample(None, stats=Stats([S.sx, S.n, S.sxx, S.n]))

    def resample_models(self, x, y, z):
        """ilr.py:173-175 without the dense one_hot table."""
        S = self._bind(*self._as2d(x, y)).label_stats(z, self.size)
        self.models.resample(None, None, stats=self._block_stats(S)[1])

    # ---- mean field --------------------------------------------------------------------------------
    def expected_log_complete_likelihood(self, x, y):
        eng = self._bind(*self._as2d(x, y))
        eng.estep(*self.canonical_expected(), stats=False, keep_logp=True)
        return eng.get_logp(self.size)

    def expected_responsibilities(self, x, y):
        eng = self._bind(*self._as2d(x, y))
        eng.estep(*self.canonical_expected(), stats=False, keep_resp=True)
        return eng.get_resp(self.size)

    def meanfield_coordinate_descent(self, x, y, randomize=True, maxiter=250, tol=1e-8,
                                     progress_bar=True, process_id=0, sample_likelihood=True, init_rng='host', seed=0):
        """ilr.py:196-228."""
        xx, yy = self._scaled(x, y)
        eng = self._bind(xx, yy)
        if randomize:
            S = random_start(eng, self.size, init_rng, seed)
        else:
            S, _ = eng.estep(*self.canonical_expected())
        vlb = []
        with tqdm(total=maxiter, desc=f'VI #{process_id + 1}', position=process_id,
                  disable=not progress_bar) as pbar:
            for _ in range(maxiter):
                S, bound = self.meanfield_iteration(eng, S, sample_likelihood)
                vlb.append(bound)
                if len(vlb) > 1 and abs(vlb[-1] - vlb[-2]) < tol:
                    return vlb
                pbar.update(1)
        return vlb

    def meanfield_iteration(self, eng, S, sample_likelihood=True):
        """One iteration of the coordinate descent (ilr.py:211-226): returns (S', ELBO).  The point-estimate draws
        the reference makes inside meanfield_update (basis, models, gating — in that order) run after the next
        pass has been launched, like the bound's prior terms: the pass only reads the posteriors."""
        self._update_from_stats(S, sample=False)
        if hasattr(eng, "estep_async"):
            eng.estep_async(*self.canonical_expected())
            if sample_likelihood:
                self._refresh_likelihoods()
            prior_terms = self._vlb_prior_terms()
            S, sc = eng.estep_wait()
        else:
            S, sc = eng.estep(*self.canonical_expected())
            if sample_likelihood:
                self._refresh_likelihoods()
            prior_terms = self._vlb_prior_terms()
        return S, prior_terms + sc[0]

    def _refresh_likelihoods(self):
        self.basis.refresh_likelihood()
        self.models.refresh_likelihood()
        self.gating.refresh_likelihood()

    def _update_from_stats(self, S, sample=True):
        bstats, mstats = self._block_stats(S)
        self.basis.meanfield_update(None, stats=bstats, sample=sample)
        self.models.meanfield_update(None, None, stats=mstats, sample=sample)
        self.gating.meanfield_update(None, S.gating_counts, sample=sample)

    def _vlb_prior_terms(self):
        return self.gating.variational_lowerbound() + np.sum(self.basis.variational_lowerbound())\
            + np.sum(self.models.variational_lowerbound())

    def meanfield_update_parameters(self, x, y, resp):
        eng = self._bind(*self._as2d(x, y))
        self._update_from_stats(eng.weighted_stats(resp))

    def meanfield_update_gating(self, resp):
        self.gating.meanfield_update(None, np.asarray(resp))

    def meanfield_update_basis(self, x, resp):
        self.basis.meanfield_update(x, resp)

    def meanfield_update_models(self, x, y, resp):
        self.models.meanfield_update(x, y, resp)

    # ---- SVI ---------------------------------------------------------------------------------------
    def meanfield_stochastic_descent(self, x, y, randomize=True, maxiter=500, step_size=1e-3, batch_size=128,
                                     progress_bar=True, procces_id=0, sample_likelihood=True):
        """ilr.py:245-277."""
        xx, yy = self._scaled(x, y)
        eng = self._bind(xx, yy)
        zz = joint_rows(xx, yy)
        if self._batch_engine is None:
            self._batch_engine = eng.spawn()      # same kind of engine (sharded stays sharded: see below)
        beng = self._batch_engine
        # fraction of the data one minibatch covers.  Sharded: every rank draws `batch_size` of ITS rows and the
        # statistics are all-reduced, so the minibatch is the union over ranks and the data set is all shards
        scale = eng.global_rows(batch_size) / float(eng.global_rows(len(xx)))

        def step(Sb):
            bstats, mstats = self._block_stats(Sb)
            self.basis.meanfield_sgd(None, None, scale, step_size, stats=bstats, sample=sample_likelihood)
            self.models.meanfield_sgd(None, None, None, scale, step_size, stats=mstats, sample=sample_likelihood)
            self.gating.meanfield_sgd(None, Sb.gating_counts, scale, step_size, sample=sample_likelihood)

        with tqdm(total=maxiter, desc=f'SVI #{procces_id + 1}', position=procces_id,
                  disable=not progress_bar) as pbar:
            return _svi.run(eng, beng, len(xx), maxiter, batch_size, randomize, self.size, lambda batch: beng.upload(zz[batch, :]),
                            self.canonical_expected, step, self._vlb_prior_terms, lambda: pbar.update(1))

    # ---- ELBO with explicit responsibilities (reference-shaped) -------------------------------------
    def variational_lowerbound_data(self, x, y, resp):
        """ilr.py:293-297: sum resp * (basis + models expected log-densities) = <Theta, S(resp)>."""
        eng = self._bind(*self._as2d(x, y))
        S = eng.weighted_stats(resp)
        cbw = embed_joint(self.basis.canonical_expected(), self.models.canonical_expected(), 0., self.input_dim)
        return canonical_inner(*cbw, S)

    def variational_lowerbound_labels(self, resp):
        resp = np.asarray(resp, dtype=float)
        nk = np.sum(resp, axis=1)
        return float(np.sum(nk * self.gating.expected_log_gating())) + self.engine.table_entropy(resp)

    def variational_lowerbound(self, x, y, resp):
        return self._vlb_prior_terms() + self.variational_lowerbound_data(x, y, resp)\
            + self.variational_lowerbound_labels(resp)

    # ---- posterior-predictive path (ilr.py:325-430) ---------------------------------------------
    def _predictive_gate(self):
        """(c, b, W) over x of  log E_q[pi_k] + log N(x; basis posterior predictive)  (ilr.py:339-345)."""
        c, b, W = self.basis.predictive_canonical()
        return c + np.log(self.gating.posterior.mean()), b, W

    @staticmethod
    def _check_dist(dist):
        if dist != 'gaussian':
            from mimo_amd.distributions.bayesian import _STUDENTT_DEFECT
            raise NotImplementedError(_STUDENTT_DEFECT)

    def meanfield_predictive_weights(self, x, dist='gaussian'):
        """(K, N) softmax_k of the gate, on the engine (x already in model coordinates)."""
        self._check_dist(dist)
        eng = _engine.bind(self.engine, np.ascontiguousarray(np.reshape(x, (-1, self.input_dim)), dtype=float))
        eng.estep(*self._predictive_gate(), stats=False, keep_resp=True)
        return eng.get_resp(self.size)

    def meanfield_predictive_activation(self, x, dist='gaussian'):
        """ilr.py:325-337 — as the weights, from raw inputs."""
        x = np.reshape(x, (-1, self.input_dim))
        return self.meanfield_predictive_weights(self.input_transform.transform(x) if self.scale else x, dist)

    def meanfield_predictive_moments(self, x, dist='gaussian'):
        """ilr.py:350-358: mus (K,N,dy), covars (K,N,dy,dy) = cs_kn (df_k psi_k)^-1."""
        self._check_dist(dist)
        x = np.reshape(x, (-1, self.input_dim))
        Ms, _, Cc, _, _ = self.models.predictive_blocks()
        xt = np.hstack((x, np.ones((len(x), 1)))) if self.affine else x
        return np.einsum('kdl,nl->knd', Ms, xt), self.models._scale_table(x)[:, :, None, None] * Cc[:, None, :, :]

    def meanfiled_log_predictive_likelihood(self, x, y, dist='gaussian'):
        """(sic, ilr.py:360-363)"""
        self._check_dist(dist)
        return self.models.log_posterior_predictive_gaussian(x, y)

    @staticmethod
    def mixture_moments(mus, covars, weights):
        """ilr.py:364-372 for caller-supplied tables (the fused path never builds them)."""
        mu = np.einsum('knd,kn->nd', mus, weights)
        covar = np.einsum('kndl,kn->ndl', covars + np.einsum('knd,knl->kndl', mus, mus), weights)\
            - np.einsum('nd,nl->ndl', mu, mu)
        return mu, covar

    def meanfield_prediction(self, x, y=None, prediction='average', dist='gaussian', incremental=False,
                             variance='diagonal'):
        """ilr.py:374-430.  One fused pass over the inputs (mimo_predict): gate softmax, expert predictive
        moments and their mixture (or the arg-max component) per row; the (K,N,.) tables of the reference
        are never materialised.  With y the negative log predictive density the reference intends at
        :405-409 is returned as well (its own call raises for stacked models — see the oracle)."""
        self._check_dist(dist)
        if prediction not in ('average', 'mode'):
            raise NotImplementedError(prediction)
        x = np.reshape(x, (-1, self.input_dim))
        if y is not None:
            y = np.reshape(y, (-1, self.output_dim))
        xx = self.input_transform.transform(x) if self.scale else x
        yy = (self.output_transform.transform(y) if self.scale else y) if y is not None else None
        Ms, Q, Cc, P, ld = self.models.predictive_blocks()
        eng = _engine.bind(self.engine, np.ascontiguousarray(xx, dtype=float))
        diag = variance == 'diagonal'          # variances and standard deviations come from the kernel: no (N, dy, dy) block over PCIe
        mu, second, nlpd = eng.predict(*self._predictive_gate(), Ms, Q, Cc, affine=self.affine, mode=prediction,
                                       y=None if yy is None else np.ascontiguousarray(yy), P=P, ld=ld,
                                       variance='diagonal' if diag else 'full')
        if diag:
            var, std = second
            if self.scale:                         # diag(S covar S') = var * sigma^2 for the diagonal output scaling S
                mu = self.output_transform.inverse_transform(mu)
                var = var * self.output_transform.var_
                std = np.sqrt(var)
            if incremental:
                mu += x[:, :self.output_dim]
            out = (mu, var, std)
        else:
            covar = second
            if self.scale:
                mu = self.output_transform.inverse_transform(mu)
                mat = np.diag(np.sqrt(self.output_transform.var_))
                covar = np.einsum('kh,...hj,ji->...ki', mat, covar, mat.T)
            if incremental:
                mu += x[:, :self.output_dim]
            var = np.diagonal(covar, axis1=1, axis2=2).copy()     # (the reference stacks N np.diag calls, ilr.py:417)
            out = (mu, covar, np.sqrt(var))
        return out + (nlpd,) if y is not None else out
