"""Hierarchical mixtures of Gaussians — same class / method surface as the
reference's mimo/mixtures/hgmm.py:

    BayesianMixtureOfGaussiansWithHierarchicalPrior   hgmm.py:118-289   K Gaussians, one shared precision
                                                                        under a Normal-Wishart hyper-prior
    MixtureOfMixtureOfGaussians                       hgmm.py:16-115    EM over M inner mixtures
    BayesianMixtureOfMixtureOfGaussians               hgmm.py:292-504   Gibbs / VI / SVI over M inner mixtures

Every pass over the data is the same fused kernel as in gmm.py.  An outer mixture hands its
responsibilities resp[m, :] to inner mixture m as per-row weights (`row_weights=` of the engine), and the
inner mixtures' per-datum log-normalisers are the rows of the outer log-density table — the (K, N) tables
of the inner mixtures never exist.
"""
import numpy as np
import numpy.random as npr
from scipy.special import logsumexp
from tqdm import tqdm

from mimo_amd import engine as _engine
from mimo_amd.distributions.bayesian import CategoricalWithDirichlet, CategoricalWithStickBreaking
from mimo_amd.mixtures.gmm import MixtureOfGaussians, _component_stats
from mimo_amd.utils.data import batches


def _softmax_rows(log_lik):
    return np.exp(log_lik - logsumexp(log_lik, axis=0, keepdims=True))


class BayesianMixtureOfGaussiansWithHierarchicalPrior:
    """reference: hgmm.py:118-289."""

    def __init__(self, size, dim, gating, components, engine=None):
        self.size = size
        self.dim = dim
        self.gating = gating
        self.components = components
        self.likelihood = MixtureOfGaussians(gating=self.gating.likelihood, components=self.components.likelihood,
                                             engine=engine)
        self._engine = engine
        self.labels_ = None

    @property
    def engine(self):
        return self._engine if self._engine is not None else self.components.likelihood.engine

    def _bind(self, obs):
        return _engine.bind(self.engine, np.asarray(obs, dtype=float).reshape(-1, self.dim),
                            self.components.likelihood.structure)

    def canonical_expected(self):
        c, b, W = self.components.canonical_expected()
        return c + self.gating.expected_log_gating(), b, W

    # ---- Gibbs sampling (hgmm.py:136-163: labels -> gating -> components) ---------------------------
    def resample(self, obs, maxiter=250, maxsubiter=5, progress_bar=True, process_id=0):
        eng = self._bind(obs)
        with tqdm(total=maxiter, desc=f'Init #{process_id + 1}', position=process_id,
                  disable=not progress_bar) as pbar:
            for _ in range(maxiter):
                u = npr.random(size=(1, eng.N))                      # stats.py:14
                labels, S = eng.gibbs_labels(*self.likelihood.canonical(), u=u)
                self.gating.resample(None, counts=S.n)
                self.components.resample(None, None, maxsubiter, stats=_component_stats(S, self.components))
                self.labels_ = labels
                pbar.update(1)

    def resample_labels(self, obs):
        eng = self._bind(obs)
        labels, _ = eng.gibbs_labels(*self.likelihood.canonical(), u=npr.random(size=(1, eng.N)), stats=False,
                                     keep_logp=True)
        return eng.get_logp(self.size), labels

    def resample_gating(self, labels):
        self.gating.resample(np.asarray(labels).astype(int))

    def resample_components(self, obs, labels, maxsubiter):
        eng = self._bind(obs)
        self.components.resample(None, None, maxsubiter, stats=_component_stats(eng.label_stats(labels, self.size), self.components))

    # ---- tables ---------------------------------------------------------------------------------------
    def expected_log_complete_likelihood(self, obs):
        eng = self._bind(obs)
        eng.estep(*self.canonical_expected(), stats=False, keep_logp=True)
        return eng.get_logp(self.size)

    def expected_log_likelihood(self, obs):
        eng = self._bind(obs)
        eng.estep(*self.canonical_expected(), stats=False, keep_lse=True)
        return eng.get_lse()

    def expected_responsibilities(self, obs):
        eng = self._bind(obs)
        eng.estep(*self.canonical_expected(), stats=False, keep_resp=True)
        return eng.get_resp(self.size)

    # ---- mean field (hgmm.py:186-228) -------------------------------------------------------------------
    def _first_stats(self, eng, randomize, weights):
        """Statistics of the initial responsibilities (random, or those of the current posterior)."""
        if randomize:
            resp = npr.rand(self.size, eng.N)
            resp /= np.sum(resp, axis=0)
            return eng.weighted_stats(resp if weights is None else resp * np.asarray(weights))
        return eng.estep(*self.canonical_expected(), row_weights=weights)[0]

    def meanfield_coordinate_descent(self, obs, randomize=True, weights=None, maxiter=250, maxsubiter=5, tol=1e-8,
                                     progress_bar=True, process_id=0):
        """One fused pass per iteration: the E-step under the new posterior yields the bound's data / label
        terms (unweighted responsibilities, hgmm.py:207) and the weighted statistics of the next update."""
        eng = self._bind(obs)
        weights = _engine.freeze_weights(weights)          # the same vector in every iteration: fingerprinted once
        S = self._first_stats(eng, randomize, weights)
        vlb = []
        with tqdm(total=maxiter, desc=f'VI #{process_id + 1}', position=process_id,
                  disable=not progress_bar) as pbar:
            for _ in range(maxiter):
                self._update_from_stats(S, maxsubiter)
                if hasattr(eng, "estep_async") and not hasattr(eng, "inner"):
                    # the pass is launched first; the bound's prior terms (K log-partition functions, the hyper-posterior's
                    # cross-entropies: 0.45 ms at K = 64, D = 16) are computed while it runs
                    eng.estep_async(*self.canonical_expected(), row_weights=weights)
                    prior_terms = self._vlb_prior_terms()
                    S, sc = eng.estep_wait()
                else:
                    S, sc = eng.estep(*self.canonical_expected(), row_weights=weights)
                    prior_terms = self._vlb_prior_terms()
                vlb.append(prior_terms + sc[0])
                if len(vlb) > 1 and abs(vlb[-1] - vlb[-2]) < tol:
                    return vlb
                pbar.update(1)
        return vlb

    def _update_from_stats(self, S, maxsubiter):
        self.components.meanfield_update(None, None, maxsubiter, stats=_component_stats(S, self.components))
        self.gating.meanfield_update(None, S.n)

    def _vlb_prior_terms(self):
        return self.gating.variational_lowerbound() + np.sum(self.components.variational_lowerbound())

    def meanfield_update_parameters(self, obs, resp, maxsubiter):
        self._update_from_stats(self._bind(obs).weighted_stats(resp), maxsubiter)

    def meanfield_update_gating(self, resp):
        self.gating.meanfield_update(None, np.asarray(resp))

    def meanfield_update_components(self, obs, resp, maxsubiter):
        self.components.meanfield_update(None, None, maxsubiter,
                                         stats=_component_stats(self._bind(obs).weighted_stats(resp), self.components))

    # ---- SVI (hgmm.py:231-270: full-data natural-gradient steps, no bound is recorded) ------------------
    def meanfield_stochastic_descent(self, obs, randomize=True, weights=None, maxiter=250, maxsubiter=5, scale=1,
                                     step_size=1e-2, progress_bar=True, procces_id=0):
        eng = self._bind(obs)
        S = self._first_stats(eng, randomize is True, weights)
        vlb = []
        with tqdm(total=maxiter, desc=f'SVI #{procces_id + 1}', position=procces_id,
                  disable=not progress_bar) as pbar:
            for i in range(maxiter):
                self._sgd_from_stats(S, maxsubiter, scale, step_size)
                if i + 1 < maxiter:          # (the reference's last E-step only produces a table it drops)
                    S, _ = eng.estep(*self.canonical_expected(), row_weights=weights)
                pbar.update(1)
        return vlb

    def _sgd_from_stats(self, S, maxsubiter, scale, step_size):
        self.components.meanfield_sgd(None, None, maxsubiter, scale, step_size, stats=_component_stats(S, self.components))
        self.gating.meanfield_sgd(None, S.n, scale, step_size)

    def meanfield_sgd_parameters(self, obs, resp, maxsubiter, scale, step_size):
        self._sgd_from_stats(self._bind(obs).weighted_stats(resp), maxsubiter, scale, step_size)

    # ---- bound with explicit responsibilities (hgmm.py:272-306) ------------------------------------------
    def variational_lowerbound_obs(self, obs, resp):
        from mimo_amd.mixtures.gmm import canonical_inner
        return canonical_inner(*self.components.canonical_expected(), self._bind(obs).weighted_stats(resp))

    def variational_lowerbound_labels(self, resp):
        resp = np.asarray(resp, dtype=float)
        return float(np.sum(np.sum(resp, axis=1) * self.gating.expected_log_gating())) + self.engine.table_entropy(resp)

    def variational_lowerbound(self, obs, resp):
        return self._vlb_prior_terms() + self.variational_lowerbound_labels(resp)\
            + self.variational_lowerbound_obs(obs, resp)


class MixtureOfMixtureOfGaussians:
    """EM over M inner MixtureOfGaussians (hgmm.py:16-115, plotting omitted)."""

    def __init__(self, cluster_size, mixture_size, dim, gating, components):
        self.cluster_size = cluster_size
        self.mixture_size = mixture_size
        self.dim = dim
        self.gating = gating
        self.components = components

    def log_complete_likelihood(self, obs):
        """Row m = log p(x_n | inner mixture m) + log pi_m: the inner mixtures' per-datum log-normalisers."""
        component_loglik = np.stack([self.components[m].log_likelihood(obs) for m in range(self.cluster_size)])
        with np.errstate(divide='ignore'):
            return component_loglik + np.log(self.gating.probs)[:, None]

    def log_likelihood(self, obs):
        return logsumexp(self.log_complete_likelihood(obs), axis=0)

    def responsibilities(self, obs):
        return _softmax_rows(self.log_complete_likelihood(obs))

    def max_likelihood(self, obs, randomize=True, maxiter=250, maxsubiter=5, progress_bar=True, process_id=0):
        if randomize:
            resp = npr.rand(self.cluster_size, len(obs))
            resp /= np.sum(resp, axis=0)
        else:
            resp = self.responsibilities(obs)
        log_lik = []
        with tqdm(total=maxiter, desc=f'EM #{process_id + 1}', position=process_id,
                  disable=not progress_bar) as pbar:
            for i in range(maxiter):
                for m in range(self.cluster_size):
                    self.components[m].max_likelihood(obs, weights=resp[m, :], randomize=randomize if i == 0 else False,
                                                      maxiter=maxsubiter, progress_bar=False)
                self.gating.max_likelihood(None, resp)
                lcl = self.log_complete_likelihood(obs)
                lse = logsumexp(lcl, axis=0)
                resp = np.exp(lcl - lse)
                log_lik.append(np.sum(lse))
                pbar.update(1)
        return log_lik


class BayesianMixtureOfMixtureOfGaussians:
    """reference: hgmm.py:292-504 (plotting omitted; the bound is not implemented there either)."""

    def __init__(self, cluster_size, mixture_size, dim, gating, components):
        self.cluster_size = cluster_size
        self.mixture_size = mixture_size
        self.dim = dim
        self.gating = gating
        self.components = components
        self.likelihood = MixtureOfMixtureOfGaussians(cluster_size, mixture_size, dim, gating=self.gating.likelihood,
                                                      components=[c.likelihood for c in self.components])

    # ---- Gibbs sampling (hgmm.py:318-353) -----------------------------------------------------------------
    def resample(self, obs, init_labels='prior', maxiter=250, maxsubiter=100, maxsubsubiter=5,
                 progress_bar=True, process_id=0):
        obs = np.asarray(obs, dtype=float).reshape(-1, self.dim)
        if init_labels == 'random':
            labels = npr.choice(self.cluster_size, size=(len(obs)))
        elif init_labels == 'prior':
            labels = self.gating.likelihood.rvs(len(obs))
        elif init_labels == 'posterior':
            _, labels = self.resample_labels(obs)
        else:
            raise ValueError(init_labels)
        with tqdm(total=maxiter, desc=f'Init #{process_id + 1}', position=process_id,
                  disable=not progress_bar) as pbar:
            for _ in range(maxiter):
                self.resample_components(obs, labels, maxsubiter, maxsubsubiter)
                self.resample_gating(labels)
                _, labels = self.resample_labels(obs)
                pbar.update(1)
        self.labels_ = labels

    def resample_labels(self, obs):
        """The outer table has only M rows: drawn on the host with the reference's single
        npr.random((1, N)) call (stats.py:8-21)."""
        log_prob = self.likelihood.log_complete_likelihood(obs)
        cum = np.exp(log_prob - logsumexp(log_prob, axis=0)).cumsum(0)
        u = npr.random(size=(1, log_prob.shape[1]))
        return log_prob, np.sum(u * cum[-1][None, :] > cum, axis=0, dtype=np.int32)

    def resample_gating(self, labels):
        self.gating.resample(np.asarray(labels).astype(int))

    def resample_components(self, obs, labels, maxsubiter, maxsubsubiter):
        for m in range(self.cluster_size):
            idx = np.where(labels == m)[0]
            self.components[m].resample(obs=obs[idx], maxiter=maxsubiter, maxsubiter=maxsubsubiter, progress_bar=False)

    # ---- mean field (hgmm.py:355-414) -----------------------------------------------------------------------
    def _gating_log(self):
        if isinstance(self.gating, (CategoricalWithDirichlet, CategoricalWithStickBreaking)):
            return self.gating.expected_log_gating()
        raise TypeError(type(self.gating))

    def expected_log_complete_likelihood(self, obs):
        component_loglik = np.stack([self.components[m].expected_log_likelihood(obs)
                                     for m in range(self.cluster_size)])
        return component_loglik + self._gating_log()[:, None]

    def expected_responsibilities(self, obs):
        return _softmax_rows(self.expected_log_complete_likelihood(obs))

    def meanfield_coordinate_descent(self, obs, randomize=True, maxiter=250, maxsubiter=5, maxsubsubiter=5,
                                     tol=1e-8, progress_bar=True, process_id=0):
        obs = np.asarray(obs, dtype=float).reshape(-1, self.dim)
        if randomize:
            resp = npr.rand(self.cluster_size, len(obs))
            resp /= np.sum(resp, axis=0)
        else:
            resp = self.expected_responsibilities(obs)
        vlb = []
        with tqdm(total=maxiter, desc=f'VI #{process_id + 1}', position=process_id,
                  disable=not progress_bar) as pbar:
            for i in range(maxiter):
                self.meanfield_update_parameters(obs, resp, maxsubiter, maxsubsubiter, randomize if i == 0 else False)
                resp = self.expected_responsibilities(obs)
                pbar.update(1)
        return vlb

    def meanfield_update_parameters(self, obs, resp, maxsubiter, maxsubsubiter, randomize):
        self.gating.meanfield_update(None, np.asarray(resp))
        for m in range(self.cluster_size):
            self.components[m].meanfield_coordinate_descent(obs=obs, randomize=randomize, weights=resp[m, :],
                                                            maxiter=maxsubiter, maxsubiter=maxsubsubiter,
                                                            progress_bar=False)

    # ---- SVI (hgmm.py:417-476) --------------------------------------------------------------------------------
    def meanfield_stochastic_descent(self, obs, randomize=True, maxiter=250, maxsubiter=5, maxsubsubiter=5,
                                     step_size=1e-2, batch_size=128, progress_bar=True, procces_id=0):
        obs = np.asarray(obs, dtype=float).reshape(-1, self.dim)
        vlb = []
        scale = batch_size / float(len(obs))
        with tqdm(total=maxiter, desc=f'SVI #{procces_id + 1}', position=procces_id,
                  disable=not progress_bar) as pbar:
            for i in range(maxiter):
                rnd = randomize if i == 0 else False
                for batch in batches(batch_size, len(obs)):
                    xb = np.ascontiguousarray(obs[batch, :])
                    if rnd is True:
                        resp = npr.rand(self.cluster_size, len(xb))
                        resp /= np.sum(resp, axis=0)
                    else:
                        resp = self.expected_responsibilities(xb)
                    self.meanfield_sgd_parameters(xb, resp, maxsubiter, maxsubsubiter, rnd, scale, step_size)
                pbar.update(1)
        return vlb

    def meanfield_sgd_parameters(self, obs, resp, maxsubiter, maxsubsubiter, randomize, scale, step_size):
        for m in range(self.cluster_size):
            self.components[m].meanfield_stochastic_descent(obs=obs, randomize=randomize, weights=resp[m, :],
                                                            maxiter=maxsubiter, maxsubiter=maxsubsubiter,
                                                            scale=scale, step_size=step_size, progress_bar=False)
        self.gating.meanfield_sgd(None, np.asarray(resp), scale, step_size)
