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
from tqdm import tqdm

from mimo_amd import engine as _engine
from mimo_amd.utils.abstraction import Statistics as Stats
from mimo_amd.distributions.lingauss import joint_rows
from mimo_amd.distributions.hierarchical import StackedAffineLinearGaussiansWithPrecision
from mimo_amd.mixtures.ilr import MixtureOfLinearGaussians, embed_joint
from mimo_amd.mixtures.gmm import canonical_inner


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
            return eng.weighted_stats(resp if weights is None else resp * weights)
        return eng.estep(*self.canonical_expected(), row_weights=weights)[0]

    def meanfield_coordinate_descent(self, x, y, randomize=True, weights=None, maxiter=250, maxsubiter=5, tol=1e-16,
                                     progress_bar=True, process_id=0, record_bound=False):
        """Returns [] like the reference (its append is commented out, hilr.py:198); `record_bound=True`
        returns the bound of every iteration instead — it comes with the pass at no extra cost."""
        eng = self._bind(x, y)
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
