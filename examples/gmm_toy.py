"""Toy Gaussian mixture: four 2-D clusters, K = 4 (BASELINE.json configs[0]; the flow of the reference's
examples/gmm/toy/gibbs_toy.py and vi_toy.py with this package's classes, text output instead of plots)."""
import argparse
import os
import sys

import numpy as np
import numpy.random as npr

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mimo_amd.distributions import (Categorical, Dirichlet, CategoricalWithDirichlet, StackedGaussiansWithPrecision,
                                    StackedNormalWisharts, StackedGaussiansWithNormalWisharts)
from mimo_amd.mixtures import MixtureOfGaussians, BayesianMixtureOfGaussians


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--method", choices=["gibbs", "vi", "em"], default="gibbs")
    ap.add_argument("--rows", type=int, default=10000)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1337)
    ap.add_argument("--init", choices=["prior", "random"], default="random", help="initial labels of the Gibbs sweeps")
    args = ap.parse_args()
    npr.seed(args.seed)
    K, D = 4, 2

    # data from a known mixture
    truth = MixtureOfGaussians(
        gating=Categorical(dim=K),
        components=StackedGaussiansWithPrecision(size=K, dim=D,
                                                 mus=np.array([[-3., 3.], [3., -3.], [5., 5.], [-5., -5.]]),
                                                 lmbdas=np.stack([s * np.eye(D) for s in (4., 3., 2., 1.)])))
    obs, labels = truth.rvs(args.rows)

    # the model: Dirichlet gating, Normal-Wishart components
    gating = CategoricalWithDirichlet(dim=K, prior=Dirichlet(dim=K, alphas=np.ones(K)))
    prior = StackedNormalWisharts(size=K, dim=D, mus=np.zeros((K, D)), kappas=1e-2 * np.ones(K),
                                  psis=np.stack(K * [np.eye(D)]), nus=(D + 1.) * np.ones(K) + 1e-8)
    model = BayesianMixtureOfGaussians(gating=gating, components=StackedGaussiansWithNormalWisharts(size=K, dim=D, prior=prior))

    if args.method == "gibbs":
        model.resample(obs, init_labels=args.init, maxiter=args.iters, progress_bar=False)
        mus = model.components.likelihood.mus
    elif args.method == "vi":
        # (the reference's vi_toy.py starts from random responsibilities and runs 1000 iterations to leave the
        # symmetric start; Gibbs sweeps first get there in a fraction of that)
        model.resample(obs, init_labels=args.init, maxiter=args.iters, progress_bar=False)
        vlb = model.meanfield_coordinate_descent(obs, randomize=False, maxiter=args.iters, tol=1e-8, progress_bar=False)
        print(f"ELBO: {vlb[0]:.3f} -> {vlb[-1]:.3f} in {len(vlb)} iterations, monotone: {bool(np.all(np.diff(vlb) > -1e-6))}")
        mus = model.components.posterior.mus
    else:
        model.resample(obs, init_labels=args.init, maxiter=args.iters, progress_bar=False)
        model.max_aposteriori(obs, randomize=False, maxiter=args.iters, progress_bar=False)
        mus = model.components.likelihood.mus
    order = np.lexsort((mus[:, 1], mus[:, 0]))
    print("estimated means (sorted):")
    print(np.round(mus[order], 2))
    print("log-likelihood per datum:", float(np.mean(model.likelihood.log_likelihood(obs))))


if __name__ == "__main__":
    main()
