"""Hierarchical mixtures of linear-Gaussian experts — the tied-activation
model of the reference's mimo/mixtures/hilr.py:79-290: K experts y | x ~ N(A x + c_k, Lambda^-1) that share ONE
slope matrix and ONE output precision, with a per-expert offset, over an input density whose K Gaussians share
one precision under a Normal-Wishart hyper-prior (examples/hilr/{gibbs,vi}_component.py).

One fused pass over the joint rows z = [x, y] per sweep, as in ilr.py: the input density, the experts and
the gating add up to one quadratic form per component, and the statistics block of that pass holds what
both conjugate blocks need (the reference re-contracts the data in every sub-iteration of both).
"""
import numpy as np
import numpy.random as npr
from scipy.special import logsumexp
from tqdm import tqdm

from mimo_amd import engine as _engine
from mimo_amd.utils.abstraction import Statistics as Stats
from mimo_amd.distributions.lingauss import joint_rows
from mimo_amd.distributions.hierarchical import StackedAffineLinearGaussiansWithPrecision
from mimo_amd.mixtures.ilr import MixtureOfLinearGaussians, Standardizer, embed_joint
from mimo_amd.mixtures.gmm import canonical_inner
from mimo_amd.utils.data import batches


class BayesianMixtureOfLinearGaussiansWithTiedActivation:
    """reference: hilr.py:79-290."""

    def __init__(self, size, input_dim, output_dim, gating, basis, models, scale=False, engine=None):
        self.size = size
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.gating = gating
        self.basis = basis          # input density: TiedGaussiansWithHierarchicalNormalWisharts
        self.models = models        # output density: TiedAffineLinearGaussiansWithMatrixNormalWisharts
        self.likelihood = MixtureOfLinearGaussians(size, input_dim, output_dim, gating=gating.likelihood,
                                                   basis=basis.likelihood, models=models.likelihood, engine=engine)
        self._engine = engine
        self.labels_ = None

    @property
    def engine(self):
        return self._engine if self._engine is not None else self.models.likelihood.engine

    def _bind(self, x, y):
        x = np.asarray(x, dtype=float).reshape(-1, self.input_dim)
        y = np.asarray(y, dtype=float).reshape(-1, self.output_dim)
        return _engine.bind(self.engine, joint_rows(x, y))

    def _split(self, S):
        """joint block -> (input-density statistics, expert statistics)."""
        dx = self.input_dim
        return (Stats([S.sx[:, :dx], S.n, S.sxx[:, :dx, :dx], S.n]),
                StackedAffineLinearGaussiansWithPrecision.block_stats(S, dx))

    def canonical_expected(self):
        """hilr.py:155-166: basis + models + gating, mean-field form, over z = [x, y]."""
        return embed_joint(self.basis.canonical_expected(), self.models.canonical_expected(),
                           self.gating.expected_log_gating(), self.input_dim)

    # ---- Gibbs sampling (hilr.py:121-149: labels -> gating -> basis -> models) ---------------------------
    def resample(self, x, y, maxiter=250, maxsubiter=5, progress_bar=True, process_id=0):
        eng = self._bind(x, y)
        with tqdm(total=maxiter, desc=f'Init #{process_id + 1}', position=process_id,
                  disable=not progress_bar) as pbar:
            for _ in range(maxiter):
                u = npr.random(size=(1, eng.N))                       # stats.py:14
                labels, S = eng.gibbs_labels(*self.likelihood.canonical(), u=u)
                bs, ms = self._split(S)
                self.gating.resample(None, counts=S.n)
                self.basis.resample(None, None, maxsubiter, stats=bs)
                self.models.resample(None, None, None, maxsubiter, stats=ms)
                self.labels_ = labels
                pbar.update(1)

    def resample_labels(self, x, y):
        eng = self._bind(x, y)
        labels, _ = eng.gibbs_labels(*self.likelihood.canonical(), u=npr.random(size=(1, eng.N)), stats=False,
                                     keep_logp=True)
        return eng.get_logp(self.size), labels

    # ---- tables ---------------------------------------------------------------------------------------------
    def expected_log_complete_likelihood(self, x, y):
        eng = self._bind(x, y)
        eng.estep(*self.canonical_expected(), stats=False, keep_logp=True)
        return eng.get_logp(self.size)

    def expected_log_likelihood(self, x, y):
        eng = self._bind(x, y)
        eng.estep(*self.canonical_expected(), stats=False, keep_lse=True)
        return eng.get_lse()

    def expected_responsibilities(self, x, y):
        eng = self._bind(x, y)
        eng.estep(*self.canonical_expected(), stats=False, keep_resp=True)
        return eng.get_resp(self.size)

    # ---- mean field (hilr.py:175-218: basis -> models -> gating; the reference records no bound) -------------
    def _first_stats(self, eng, randomize, weights):
        if randomize:
            resp = npr.rand(self.size, eng.N)
            resp /= np.sum(resp, axis=0)
            return eng.weighted_stats(resp if weights is None else resp * np.asarray(weights))
        return eng.estep(*self.canonical_expected(), row_weights=weights)[0]

    def meanfield_coordinate_descent(self, x, y, randomize=True, weights=None, maxiter=250, maxsubiter=5, tol=1e-16,
                                     progress_bar=True, process_id=0, record_bound=False):
        """Returns [] like the reference (its append is commented out, hilr.py:198); `record_bound=True`
        returns the bound of every iteration instead — it comes with the pass at no extra cost."""
        eng = self._bind(x, y)
        weights = _engine.freeze_weights(weights)          # the same vector in every iteration: fingerprinted once
        S = self._first_stats(eng, randomize, weights)
        vlb = []
        with tqdm(total=maxiter, desc=f'VI #{process_id + 1}', position=process_id,
                  disable=not progress_bar) as pbar:
            for i in range(maxiter):
                self._update_from_stats(S, maxsubiter)
                if i + 1 < maxiter or record_bound:
                    S, sc = eng.estep(*self.canonical_expected(), row_weights=weights)
                    if record_bound:
                        vlb.append(self._vlb_prior_terms() + sc[0])
                else:      # the reference's last E-step produces a table it drops — and the draws of reference_draw
                    self.models.reference_draw()
                pbar.update(1)
        return vlb

    def _update_from_stats(self, S, maxsubiter):
        bs, ms = self._split(S)
        self.basis.meanfield_update(None, None, maxsubiter, stats=bs)
        self.models.meanfield_update(None, None, None, maxsubiter, stats=ms)
        self.gating.meanfield_update(None, S.n)

    def meanfield_update_parameters(self, x, y, resp, maxsubiter):
        self._update_from_stats(self._bind(x, y).weighted_stats(resp), maxsubiter)

    def _vlb_prior_terms(self):
        return self.gating.variational_lowerbound() + np.sum(self.basis.variational_lowerbound())\
            + np.sum(self.models.variational_lowerbound())

    # ---- SVI (hilr.py:221-259): the experts have no stochastic update in the reference ------------------------
    def meanfield_stochastic_descent(self, x, y, randomize=True, weights=None, maxiter=250, maxsubiter=5, scale=1,
                                     step_size=1e-2, progress_bar=True, procces_id=0):
        eng = self._bind(x, y)
        S = self._first_stats(eng, randomize is True, weights)
        bs, ms = self._split(S)
        self.basis.meanfield_sgd(None, None, maxsubiter, scale, step_size, stats=bs)     # applied, as in the reference,
        self.models.meanfield_sgd(None, None, None, maxsubiter, scale, step_size, stats=ms)   # before this raises

    # ---- bound with explicit responsibilities (hilr.py:261-290) -------------------------------------------------
    def variational_lowerbound_data(self, x, y, resp):
        S = self._bind(x, y).weighted_stats(resp)
        c, b, W = embed_joint(self.basis.canonical_expected(), self.models.canonical_expected(),
                              np.zeros(self.size), self.input_dim)
        return canonical_inner(c, b, W, S)

    def variational_lowerbound_labels(self, resp):
        resp = np.asarray(resp, dtype=float)
        return float(np.sum(np.sum(resp, axis=1) * self.gating.expected_log_gating())) + self.engine.table_entropy(resp)

    def variational_lowerbound(self, x, y, resp):
        return self._vlb_prior_terms() + self.variational_lowerbound_labels(resp)\
            + self.variational_lowerbound_data(x, y, resp)


class MixtureOfMixtureOfLinearGaussians:
    """Outer point-estimate mixture over M inner MixtureOfLinearGaussians (hilr.py:18-76)."""

    def __init__(self, cluster_size, mixture_size, input_dim, output_dim, gating, components):
        self.cluster_size = cluster_size
        self.mixture_size = mixture_size
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.gating = gating
        self.components = components

    def log_complete_likelihood(self, x, y):
        rows = np.stack([self.components[m].log_likelihood(x, y) for m in range(self.cluster_size)])
        with np.errstate(divide='ignore'):
            return rows + np.log(self.gating.probs)[:, None]

    def log_likelihood(self, x, y):
        return logsumexp(self.log_complete_likelihood(x, y), axis=0)

    def responsibilities(self, x, y):
        lcl = self.log_complete_likelihood(x, y)
        return np.exp(lcl - logsumexp(lcl, axis=0, keepdims=True))

    def max_likelihood(self, x, y, randomize=True, maxiter=250, maxsubiter=5, progress_bar=True, process_id=0):
        raise NotImplementedError        # hilr.py:73-76


class BayesianMixtureOfMixtureOfLinearGaussians:
    """reference: hilr.py:293-609.  The outer table has M rows (inner log-normalisers from the engine); inner
    mixture m runs the fused pass with the outer responsibilities resp[m, :] as row weights; prediction is ONE
    mimo_predict over the M*K (cluster, expert) pairs."""

    def __init__(self, cluster_size, mixture_size, input_dim, output_dim, gating, components, scale=False):
        self.cluster_size = cluster_size
        self.mixture_size = mixture_size
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.gating = gating
        self.components = components
        self.likelihood = MixtureOfMixtureOfLinearGaussians(cluster_size, mixture_size, input_dim, output_dim,
                                                            gating=self.gating.likelihood,
                                                            components=[c.likelihood for c in self.components])
        self.scale = scale
        self.input_transform = Standardizer()
        self.output_transform = Standardizer()
        self.labels_ = None

    @property
    def engine(self):
        return self.components[0].engine

    def init_transform(self, x, y):
        self.scale = True
        self.input_transform.fit(x)
        self.output_transform.fit(y)

    def _scaled(self, x, y):
        x = np.reshape(np.asarray(x, dtype=float), (-1, self.input_dim))
        y = np.reshape(np.asarray(y, dtype=float), (-1, self.output_dim))
        if self.scale:
            return np.ascontiguousarray(self.input_transform.transform(x)), np.ascontiguousarray(self.output_transform.transform(y))
        return x, y

    def used_labels(self, x, y):
        z = np.argmax(self.expected_responsibilities(*self._scaled(x, y)), axis=0)
        return np.where(np.bincount(z, minlength=self.cluster_size) > 0)[0]

    def max_aposteriori(self, x, y, randomize=True, maxiter=250, maxsubiter=5, progress_bar=True, process_id=0):
        raise NotImplementedError        # hilr.py:340-343

    # ---- Gibbs sampling (hilr.py:346-387) ----------------------------------------------------------------------
    def resample(self, x, y, init_labels='prior', maxiter=250, maxsubiter=100, maxsubsubiter=5,
                 progress_bar=True, process_id=0):
        xx, yy = self._scaled(x, y)
        if init_labels == 'random':
            z = npr.choice(self.cluster_size, size=(len(xx)))
        elif init_labels == 'posterior':
            _, z = self.resample_labels(xx, yy)
        elif init_labels == 'prior':
            z = self.gating.likelihood.rvs(len(xx))
        else:
            raise ValueError(init_labels)
        with tqdm(total=maxiter, desc=f'Init #{process_id + 1}', position=process_id,
                  disable=not progress_bar) as pbar:
            for _ in range(maxiter):
                for m in range(self.cluster_size):
                    idx = np.where(z == m)[0]
                    self.components[m].resample(np.ascontiguousarray(xx[idx]), np.ascontiguousarray(yy[idx]),
                                                maxiter=maxsubiter, maxsubiter=maxsubsubiter, progress_bar=False)
                self.gating.resample(np.asarray(z).astype(int))
                _, z = self.resample_labels(xx, yy)
                pbar.update(1)
        self.labels_ = z

    def resample_labels(self, x, y):
        log_prob = self.likelihood.log_complete_likelihood(x, y)
        cum = np.exp(log_prob - logsumexp(log_prob, axis=0)).cumsum(0)
        u = npr.random(size=(1, log_prob.shape[1]))                    # stats.py:14
        return log_prob, np.sum(u * cum[-1][None, :] > cum, axis=0, dtype=np.int32)

    # ---- mean field (hilr.py:389-458) ----------------------------------------------------------------------------
    def expected_log_complete_likelihood(self, x, y):
        rows = np.stack([self.components[m].expected_log_likelihood(x, y) for m in range(self.cluster_size)])
        return rows + self.gating.expected_log_gating()[:, None]

    def expected_responsibilities(self, x, y):
        lcl = self.expected_log_complete_likelihood(x, y)
        return np.exp(lcl - logsumexp(lcl, axis=0, keepdims=True))

    def meanfield_coordinate_descent(self, x, y, randomize=True, maxiter=250, maxsubiter=5, maxsubsubiter=5,
                                     tol=1e-16, progress_bar=True, process_id=0):
        xx, yy = self._scaled(x, y)
        if randomize:
            resp = npr.rand(self.cluster_size, len(xx))
            resp /= np.sum(resp, axis=0)
        else:
            resp = self.expected_responsibilities(xx, yy)
        vlb = []
        with tqdm(total=maxiter, desc=f'VI #{process_id + 1}', position=process_id,
                  disable=not progress_bar) as pbar:
            for i in range(maxiter):
                self.meanfield_update_parameters(xx, yy, resp, maxsubiter, maxsubsubiter, randomize if i == 0 else False)
                resp = self.expected_responsibilities(xx, yy)
                pbar.update(1)
        return vlb

    def meanfield_update_parameters(self, x, y, resp, maxsubiter, maxsubsubiter, randomize):
        for m in range(self.cluster_size):
            self.components[m].meanfield_coordinate_descent(x, y, randomize=randomize, weights=resp[m, :],
                                                            maxiter=maxsubiter, maxsubiter=maxsubsubiter,
                                                            progress_bar=False)
        self.gating.meanfield_update(None, np.asarray(resp))

    # ---- SVI (hilr.py:460-514): reaches the experts' missing stochastic update and raises ------------------------
    def meanfield_stochastic_descent(self, x, y, randomize=True, maxiter=250, maxsubiter=5, maxsubsubiter=5,
                                     step_size=1e-2, batch_size=128, progress_bar=True, procces_id=0):
        xx, yy = self._scaled(x, y)
        scale = batch_size / float(len(xx))
        for i in range(maxiter):
            rnd = randomize if i == 0 else False
            for batch in batches(batch_size, len(xx)):
                xb, yb = np.ascontiguousarray(xx[batch, :]), np.ascontiguousarray(yy[batch, :])
                if rnd is True:
                    resp = npr.rand(self.cluster_size, len(xb))
                    resp /= np.sum(resp, axis=0)
                else:
                    resp = self.expected_responsibilities(xb, yb)
                for m in range(self.cluster_size):
                    self.components[m].meanfield_stochastic_descent(xb, yb, randomize=rnd, weights=resp[m, :],
                                                                    maxiter=maxsubiter, maxsubiter=maxsubsubiter,
                                                                    scale=scale, step_size=step_size, progress_bar=False)
                self.gating.meanfield_sgd(None, np.asarray(resp), scale, step_size)
        return []

    # ---- posterior-predictive path (hilr.py:527-609) -------------------------------------------------------------
    def _flat_gate(self):
        """(c, b, W) over x for the M*K pairs: log E[pi_m] + log E[pi_k|m] + log N(x; basis predictive of (m, k))."""
        cs, bs, Ws = [], [], []
        lg = np.log(self.gating.posterior.mean())
        for m, comp in enumerate(self.components):
            c, b, W = comp.basis.predictive_canonical()
            cs.append(c + np.log(comp.gating.posterior.mean()) + lg[m]); bs.append(b); Ws.append(W)
        return np.concatenate(cs), np.concatenate(bs), np.concatenate(Ws)

    def meanfield_predictive_weights(self, x):
        """(M, K, N), normalised jointly over (m, k) (hilr.py:542-551; x in model coordinates)."""
        x = np.ascontiguousarray(np.reshape(x, (-1, self.input_dim)), dtype=float)
        eng = _engine.bind(self.engine, x)
        eng.estep(*self._flat_gate(), stats=False, keep_resp=True)
        return eng.get_resp(self.cluster_size * self.mixture_size).reshape(self.cluster_size, self.mixture_size, -1)

    def meanfield_predictive_activation(self, x):
        x = np.reshape(x, (-1, self.input_dim))
        return self.meanfield_predictive_weights(self.input_transform.transform(x) if self.scale else x)

    def meanfield_predictive_moments(self, x):
        """mus (M, K, N, dy), covars (M, K, N, dy, dy) (hilr.py:553-560)."""
        out = [comp.models.posterior_predictive_gaussian(x) for comp in self.components]
        return np.stack([o[0] for o in out]), np.linalg.inv(np.stack([o[1] for o in out]))

    @staticmethod
    def mixture_moments(mus, covars, weights):
        mean = np.einsum('mknd,mkn->nd', mus, weights)
        covar = np.einsum('mkndl,mkn->ndl', covars + np.einsum('mknd,mknl->mkndl', mus, mus), weights)\
            - np.einsum('nd,nl->ndl', mean, mean)
        return mean, covar

    def meanfield_prediction(self, x, prediction='average', incremental=False, variance='diagonal'):
        if prediction not in ('average', 'mode'):
            raise NotImplementedError(prediction)
        x = np.reshape(x, (-1, self.input_dim))
        xx = self.input_transform.transform(x) if self.scale else x
        blocks = [comp.models.predictive_blocks() for comp in self.components]
        Ms, Q, Cc = (np.concatenate([b[i] for b in blocks]) for i in range(3))
        eng = _engine.bind(self.engine, np.ascontiguousarray(xx, dtype=float))
        mean, covar, _ = eng.predict(*self._flat_gate(), Ms, Q, Cc, affine=True, mode=prediction)
        if self.scale:
            mean = self.output_transform.inverse_transform(mean)
            mat = np.diag(np.sqrt(self.output_transform.var_))
            covar = np.einsum('kh,...hj,ji->...ki', mat, covar, mat.T)
        if incremental:
            mean += x[:, :self.output_dim]
        var = np.diagonal(covar, axis1=1, axis2=2).copy()
        return (mean, var if variance == 'diagonal' else covar, np.sqrt(var))
