"""(Bayesian) mixtures of Gaussians — drivers of EM, MAP, Gibbs sampling, mean-field VI and SVI.

Same class / method surface as the reference's mimo/mixtures/gmm.py, re-built around ONE fused
pass over the device-resident data per sweep (HipEngine.estep / HipEngine.gibbs_labels):

    reference VI iteration (gmm.py:275-285)           this module
    --------------------------------------------      ------------------------------------------
    weighted_statistics(obs, resp)   O(NKD^2)  \
    expected_responsibilities(obs)   O(NKD^2)   }     engine.estep(c, b, W) -> S, sum_n lse_n
    variational_lowerbound_obs(...)  O(NKD^2)  /      (responsibilities never leave the GPU)
    conjugate update + ELBO prior terms  O(KD^3)      unchanged (host, float64 NumPy)

The data/label ELBO terms come from the identity (verified on the reference)
    sum resp * E[log p(x|k)] + sum resp * E[log pi] - sum resp log resp = sum_n logsumexp_k l[k,n].
Host RNG (Wishart / Dirichlet / Beta draws, optional label uniforms) stays numpy.random in the
reference's call order, so seeded runs are comparable.
"""
import numpy as np
import numpy.random as npr
from tqdm import tqdm

from mimo_amd import engine as _engine
from mimo_amd.distributions import native_sweep as _native_sweep
from mimo_amd.mixtures import _svi
from mimo_amd.utils.abstraction import Statistics as Stats
from mimo_amd.utils.data import batches


def _component_stats(S, components=None):
    """engine block -> the Stats tuple of the component family: Stats([sum r x, n, sum r xx', n])
    (gaussian.py:502) or, for diagonal precisions, Stats([sum r x, n_d, n_d, sum r x^2]) (gaussian.py:815)."""
    lik = getattr(components, 'likelihood', components)
    if lik is not None and hasattr(lik, 'block_stats'):
        return lik.block_stats(S)
    return Stats([S.sx, S.n, S.sxx, S.n])


def random_start(eng, K, init_rng='host', seed=0):
    """Statistics of the drivers' random start (gmm.py:265-267: resp = rand(K, N); resp /= resp.sum(0)).
    init_rng='host'  : numpy.random.rand(K, N) exactly as the reference draws it (seeded parity) — K N doubles over PCIe;
    init_rng='philox': the table is drawn on the device from the Philox stream keyed by `seed`, counter (row, k)
                       (HipEngine.random_resp_stats): no host draw, no upload — 5 GB less traffic at N = 1e7, K = 64."""
    if init_rng == 'philox':
        return eng.random_resp_stats(K, seed)
    if init_rng != 'host':
        raise ValueError(init_rng)
    resp = npr.rand(K, eng.N)
    resp /= np.sum(resp, axis=0)
    return eng.weighted_stats(resp)


def canonical_inner(c, b, W, S):
    """sum_kn r_kn l_kn = <Theta, S(r)>  for l = c + b.x - 1/2 x'Wx  and S the statistics of r."""
    if S.sxx is None:      # 'linear' structure: one W for all components, pooled second moment
        return float(np.sum(c * S.n) + np.sum(b * S.sx) - 0.5 * np.sum(W[0] * S.sxx_total))
    return float(np.sum(c * S.n) + np.sum(b * S.sx) - 0.5 * np.sum(W * S.sxx))


class LazyTable:
    """A (K, N) table that is computed on first use (np.asarray(t), t[...], arithmetic through __array__)."""

    def __init__(self, compute, shape):
        self._compute, self._value, self.shape = compute, None, tuple(shape)

    def __array__(self, dtype=None, copy=None):
        if self._value is None:
            self._value = self._compute()
        return self._value if dtype is None else self._value.astype(dtype, copy=False)

    def __getitem__(self, idx):
        return self.__array__()[idx]

    def __len__(self):
        return self.shape[0]

    @property
    def evaluated(self):
        return self._value is not None


class MixtureOfGaussians:
    """reference: mimo/mixtures/gmm.py:16-145 (plotting omitted)."""

    def __init__(self, gating, components, engine=None):
        assert components.size == gating.dim
        self.gating = gating
        self.components = components
        self._engine = engine

    @property
    def engine(self):
        return self._engine if self._engine is not None else self.components.engine

    @property
    def size(self):
        return self.gating.dim

    @property
    def dim(self):
        return self.components.dim

    def used_labels(self, obs):
        labels = np.argmax(self.responsibilities(obs), axis=0)
        return np.where(np.bincount(labels, minlength=self.size) > 0)[0]

    def rvs(self, size=1):
        """gmm.py:50-60 (same RNG order: labels, per-component draws, permutation)."""
        labels = self.gating.rvs(size)
        counts = np.bincount(labels, minlength=self.size)
        obs = np.zeros((size, self.dim))
        ci = self.components.lmbdas_chol_inv
        for idx, count in enumerate(counts):
            shape = self.dim if count == 1 else (count, self.dim)
            obs[labels == idx, ...] = self.components.mus[idx] + npr.normal(size=shape).dot(ci[idx].T)
        perm = npr.permutation(size)
        return obs[perm], labels[perm]

    # ---- canonical form ---------------------------------------------------------------------
    def canonical(self):
        c, b, W = self.components.canonical()
        with np.errstate(divide='ignore'):
            return c + np.log(self.gating.probs), b, W

    def _bind(self, obs):
        return _engine.bind(self.engine, np.asarray(obs, dtype=float).reshape(-1, self.dim),
                            getattr(self.components, 'structure', 'full'))

    # ---- reference-shaped table methods --------------------------------------------------------
    def log_complete_likelihood(self, obs):
        eng = self._bind(obs)
        eng.estep(*self.canonical(), stats=False, keep_logp=True)
        return eng.get_logp(self.size)

    def log_likelihood(self, obs):
        eng = self._bind(obs)
        eng.estep(*self.canonical(), stats=False, keep_lse=True)
        return eng.get_lse()

    def responsibilities(self, obs):
        eng = self._bind(obs)
        eng.estep(*self.canonical(), stats=False, keep_resp=True)
        return eng.get_resp(self.size)

    # ---- EM ------------------------------------------------------------------------------------
    def max_likelihood(self, obs, randomize=True, weights=None, maxiter=250, progress_bar=True, process_id=0,
                       init_rng='host', seed=0):
        """gmm.py:77-103.  Each iteration is one fused pass: the E-step under the new parameters
        also yields the statistics of the next M-step and sum_n log p(x_n)."""
        eng = self._bind(obs)
        if weights is not None:
            return self._max_likelihood_weighted(eng, randomize, weights, maxiter, progress_bar, process_id)
        if randomize:
            S = random_start(eng, self.size, init_rng, seed)
        else:
            S, _ = eng.estep(*self.canonical())
        log_lik = []
        with tqdm(total=maxiter, desc=f'EM #{process_id + 1}', position=process_id,
                  disable=not progress_bar) as pbar:
            for _ in range(maxiter):
                self.components.max_likelihood(None, stats=_component_stats(S, self.components))
                self.gating.max_likelihood(None, S.gating_counts)
                S, sc = eng.estep(*self.canonical())
                log_lik.append(sc[0])
                pbar.update(1)
        return log_lik

    def _max_likelihood_weighted(self, eng, randomize, weights, maxiter, progress_bar, process_id):
        """per-datum weights (gmm.py:93): resp * weights needs the table; kept for API parity."""
        if randomize:
            resp = npr.rand(self.size, eng.N)
            resp /= np.sum(resp, axis=0)
        else:
            eng.estep(*self.canonical(), stats=False, keep_resp=True)
            resp = eng.get_resp(self.size)
        log_lik = []
        with tqdm(total=maxiter, desc=f'EM #{process_id + 1}', position=process_id,
                  disable=not progress_bar) as pbar:
            for _ in range(maxiter):
                S = eng.weighted_stats(resp * weights)
                self.components.max_likelihood(None, stats=_component_stats(S, self.components))
                self.gating.max_likelihood(None, S.gating_counts)
                _, sc = eng.estep(*self.canonical(), stats=False, keep_resp=True)
                resp = eng.get_resp(self.size)
                log_lik.append(sc[0])
                pbar.update(1)
        return log_lik


class BayesianMixtureOfGaussians:
    """reference: mimo/mixtures/gmm.py:147-371 (plotting omitted)."""

    def __init__(self, gating, components, engine=None):
        self.gating = gating
        self.components = components
        self.likelihood = MixtureOfGaussians(gating=self.gating.likelihood,
                                             components=self.components.likelihood, engine=engine)
        self._engine = engine
        self._batch_engine = None
        self.labels_ = None

    @property
    def engine(self):
        return self._engine if self._engine is not None else self.components.likelihood.engine

    @property
    def size(self):
        return self.likelihood.size

    @property
    def dim(self):
        return self.likelihood.dim

    def _structure(self):
        return getattr(self.components.likelihood, 'structure', 'full')

    def _bind(self, obs):
        return _engine.bind(self.engine, np.asarray(obs, dtype=float).reshape(-1, self.dim), self._structure())

    def used_labels(self, obs):
        labels = np.argmax(self.expected_responsibilities(obs), axis=0)
        return np.where(np.bincount(labels, minlength=self.size) > 0)[0]

    # ---- canonical forms ---------------------------------------------------------------------
    def canonical_expected(self):
        """VI form: <E_q[eta_k], t(x)> + E[log pi_k]  (gmm.py:244-254)."""
        c, b, W = self.components.canonical_expected()
        return c + self.gating.expected_log_gating(), b, W

    # ---- MAP-EM --------------------------------------------------------------------------------
    def max_aposteriori(self, obs, randomize=True, maxiter=250, progress_bar=True, process_id=0, init_rng='host', seed=0):
        """gmm.py:176-204."""
        eng = self._bind(obs)
        if randomize:
            S = random_start(eng, self.size, init_rng, seed)
        else:
            S, _ = eng.estep(*self.likelihood.canonical())
        log_prob = []
        with tqdm(total=maxiter, desc=f'MAP #{process_id + 1}', position=process_id,
                  disable=not progress_bar) as pbar:
            for _ in range(maxiter):
                self.components.max_aposteriori(None, stats=_component_stats(S, self.components))
                self.gating.max_aposteriori(None, S.gating_counts)
                S, sc = eng.estep(*self.likelihood.canonical())
                log_prior = self.gating.prior.log_likelihood(self.gating.likelihood.params)\
                    + np.sum(self.components.prior.log_likelihood(self.components.likelihood.params))
                log_prob.append(sc[0] + log_prior)
                pbar.update(1)
        return log_prob

    # ---- Gibbs sampling ------------------------------------------------------------------------
    def resample(self, obs, init_labels='prior', maxiter=1, progress_bar=True, process_id=0,
                 label_rng='host', seed=0, param_rng=None):
        """gmm.py:207-225 — sweep order components -> gating -> labels.

        label_rng='host'   : the uniforms are numpy.random.random((1, N)) exactly as in
                             mimo/utils/stats.py:14 (seeded runs reproduce the reference's labels).
        label_rng='philox' : per-datum counter-based Philox4x32-10 inside the kernel, keyed by
                             `seed`, counter (global row, sweep) — no PCIe traffic per sweep.
        param_rng          : None keeps the reference's per-component numpy.random call order for the
                             Wishart / Gaussian draws; a numpy Generator batches the K draws.
        The label kernel also returns the statistics of the labels it drew, which are exactly what
        the next sweep's resample_components / resample_gating need: one pass per sweep."""
        eng = self._bind(obs)
        N = eng.N
        if init_labels == 'random':
            labels = npr.choice(self.size, size=(N))
        elif init_labels == 'prior':
            labels = self.gating.likelihood.rvs(N)
        elif init_labels == 'posterior':
            labels = self._draw_labels(eng, label_rng, seed, 0, stats=False)[0]
        else:
            raise ValueError(init_labels)
        S = eng.label_stats(labels, self.size)

        with tqdm(total=maxiter, desc=f'Init #{process_id + 1}', position=process_id,
                  disable=not progress_bar) as pbar:
            for it in range(maxiter):
                last = it == maxiter - 1
                labels, S = self.gibbs_iteration(eng, S, it + 1, label_rng, seed, param_rng, stats=not last,
                                                 return_labels=last)
                pbar.update(1)
        self.labels_ = labels

    resample_model = resample      # pybasicbayes-era name used by BASELINE.json's north star

    def gibbs_iteration(self, eng, S, sweep, label_rng='host', seed=0, param_rng=None, stats=True,
                        return_labels=False):
        """One sweep of `resample` (gmm.py:217-223): components and gating from the statistics `S` of the labels
        drawn last, then the label pass.  Returns (labels | None, S' | None)."""
        self.components.resample(None, stats=_component_stats(S, self.components), rng=param_rng)
        self.gating.resample(None, counts=S.gating_counts)
        return self._draw_labels(eng, label_rng, seed, sweep, stats=stats, return_labels=return_labels)

    def _draw_labels(self, eng, label_rng, seed, sweep, stats=True, return_labels=True):
        c, b, W = self.likelihood.canonical()
        if hasattr(eng, 'check_replicated_once'):      # sharded: every rank drew these blocks from ITS host generator
            eng.check_replicated_once(c, b, W, what="component / gating parameters drawn for the label pass")
        if label_rng == 'host':
            u = npr.random(size=(1, eng.N))
            return eng.gibbs_labels(c, b, W, u=u, stats=stats, return_labels=return_labels)
        if label_rng == 'philox':
            return eng.gibbs_labels(c, b, W, seed=seed, sweep=sweep, stats=stats, return_labels=return_labels)
        raise ValueError(label_rng)

    def resample_labels(self, obs, lazy=True):
        """gmm.py:227-230 -> (log_prob (K, N), labels int32).  The reference returns the table it drew from; almost
        no caller reads it, and at C3 it is 20 GB.  lazy=True (default) returns a LazyTable: the draw runs fused (no
        table anywhere), and the table is evaluated — with the parameters captured now — only if it is converted to
        an array or indexed; lazy=False computes and copies it eagerly like the reference."""
        eng = self._bind(obs)
        c, b, W = self.likelihood.canonical()
        u = npr.random(size=(1, eng.N))
        if not lazy:
            labels, _ = eng.gibbs_labels(c, b, W, u=u, stats=False, keep_logp=True)
            return eng.get_logp(self.size), labels
        labels, _ = eng.gibbs_labels(c, b, W, u=u, stats=False)
        c, b, W = np.array(c), np.array(b), np.array(W)

        def table():
            e = self._bind(obs)
            e.estep(c, b, W, stats=False, keep_logp=True)
            return e.get_logp(len(c))
        return LazyTable(table, (len(c), eng.N)), labels

    def resample_gating(self, labels):
        self.gating.resample(np.asarray(labels).astype(int))

    def resample_components(self, obs, labels):
        """gmm.py:235-237 without the dense one_hot table."""
        eng = self._bind(obs)
        self.components.resample(None, stats=_component_stats(eng.label_stats(labels, self.size), self.components))

    # ---- mean field ----------------------------------------------------------------------------
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

    def meanfield_coordinate_descent(self, obs, randomize=True, maxiter=250, tol=1e-8,
                                     progress_bar=True, process_id=0, sample_likelihood=True, init_rng='host', seed=0):
        """gmm.py:261-287.  Returns the ELBO list.  `sample_likelihood=False` drops the reference's
        per-iteration likelihood.params = posterior.rvs() (host RNG only; no effect on the ELBO)."""
        eng = self._bind(obs)
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
        """One iteration of the coordinate descent (gmm.py:275-285): conjugate update from the statistics `S` of
        the previous data pass, the next fused pass, the bound.  Returns (S', ELBO).

        The reference draws likelihood.params = posterior.rvs() inside meanfield_update (bayesian.py:225-230): K
        Bartlett draws, 2.6 ms at K = 64, D = 16.  The next E-step reads the posterior only, so the pass is launched
        first and the draws (same numpy.random call order: components, then gating) and the bound's prior terms run
        on the host while the kernel is in flight."""
        fused = _native_sweep.gmm_vi_sweep(self.gating, self.components, _component_stats(S, self.components),
                                           S.gating_counts)
        if fused is not None:           # update + canonical form in one native call, the bound's prior terms in a second
            canon, bound = fused         # one, issued once the pass is in flight
        else:
            self._update_from_stats(S, sample=False)
            canon = self.canonical_expected()
        if hasattr(eng, "estep_async"):
            eng.estep_async(*canon)
            if sample_likelihood:
                self._refresh_likelihoods()
            prior_terms = self._vlb_prior_terms() if fused is None else bound()
            S, sc = eng.estep_wait()
        else:
            S, sc = eng.estep(*canon)
            if sample_likelihood:
                self._refresh_likelihoods()
            prior_terms = self._vlb_prior_terms() if fused is None else bound()
        return S, prior_terms + sc[0]

    def _refresh_likelihoods(self):
        self.components.refresh_likelihood()
        self.gating.refresh_likelihood()

    def _update_from_stats(self, S, sample=True):
        self.components.meanfield_update(None, stats=_component_stats(S, self.components), sample=sample)
        self.gating.meanfield_update(None, S.gating_counts, sample=sample)

    def _vlb_prior_terms(self):
        return self.gating.variational_lowerbound() + np.sum(self.components.variational_lowerbound())

    def meanfield_update_parameters(self, obs, resp):
        eng = self._bind(obs)
        self._update_from_stats(eng.weighted_stats(resp))

    def meanfield_update_gating(self, resp):
        self.gating.meanfield_update(None, np.asarray(resp))

    def meanfield_update_components(self, obs, resp):
        eng = self._bind(obs)
        self.components.meanfield_update(None, stats=_component_stats(eng.weighted_stats(resp), self.components))

    # ---- SVI -----------------------------------------------------------------------------------
    def meanfield_stochastic_descent(self, obs, randomize=True, maxiter=500, step_size=1e-2, batch_size=128,
                                     progress_bar=True, procces_id=0, sample_likelihood=True):
        """gmm.py:300-326: one minibatch natural-gradient step + one full-data E-step / ELBO per
        outer iteration (batches() yields a single batch — SURVEY.md Appendix B #6)."""
        obs = np.asarray(obs, dtype=float).reshape(-1, self.dim)
        eng = self._bind(obs)
        if self._batch_engine is None:
            self._batch_engine = eng.spawn()      # same kind of engine (sharded stays sharded: see below)
        beng = self._batch_engine
        # fraction of the data one minibatch covers.  Sharded: every rank draws `batch_size` of ITS rows and the
        # statistics are all-reduced, so the minibatch is the union over ranks and the data set is all shards
        scale = eng.global_rows(batch_size) / float(eng.global_rows(len(obs)))

        def upload(batch):
            if hasattr(beng, 'set_structure'):
                beng.set_structure(self._structure())
            beng.upload(obs[batch, :])

        def step(Sb):
            self.components.meanfield_sgd(None, None, scale, step_size, stats=_component_stats(Sb, self.components),
                                          sample=sample_likelihood)
            self.gating.meanfield_sgd(None, Sb.gating_counts, scale, step_size, sample=sample_likelihood)

        with tqdm(total=maxiter, desc=f'SVI #{procces_id + 1}', position=procces_id,
                  disable=not progress_bar) as pbar:
            return _svi.run(eng, beng, len(obs), maxiter, batch_size, randomize, self.size, upload, self.canonical_expected,
                            step, self._vlb_prior_terms, lambda: pbar.update(1))

    def meanfield_sgd_parameters(self, obs, resp, scale, step_size):
        eng = self._bind(obs)
        S = eng.weighted_stats(resp)
        self.components.meanfield_sgd(None, None, scale, step_size, stats=_component_stats(S, self.components))
        self.gating.meanfield_sgd(None, S.gating_counts, scale, step_size)

    # ---- ELBO with explicit responsibilities (reference-shaped) ----------------------------------
    def variational_lowerbound_obs(self, obs, resp):
        """gmm.py:338-339: sum resp * E[log p(x|k)] = <Theta_components, S(resp)> (no second table)."""
        eng = self._bind(obs)
        return canonical_inner(*self.components.canonical_expected(), eng.weighted_stats(resp))

    def variational_lowerbound_labels(self, resp):
        """gmm.py:341-356."""
        resp = np.asarray(resp, dtype=float)
        nk = np.sum(resp, axis=1)
        vlb = float(np.sum(nk * self.gating.expected_log_gating()))
        return vlb + self.engine.table_entropy(resp)

    def variational_lowerbound(self, obs, resp):
        """gmm.py:358-364."""
        return self._vlb_prior_terms() + self.variational_lowerbound_obs(obs, resp)\
            + self.variational_lowerbound_labels(resp)
