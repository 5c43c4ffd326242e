"""Mixture of linear-Gaussian experts on a noisy sine (the flow of the reference's examples/ilr/evaluate_sine.py in
its plain form: stick-breaking gating, Gaussian basis over x, affine experts y | x; mean-field VI, then the
posterior-predictive mean / standard deviation on a grid through the fused prediction kernel)."""
import argparse
import os
import sys

import numpy as np
import numpy.random as npr

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mimo_amd.distributions import (TruncatedStickBreaking, CategoricalWithStickBreaking, StackedNormalWisharts,
                                    StackedGaussiansWithNormalWisharts, StackedMatrixNormalWisharts,
                                    StackedLinearGaussiansWithMatrixNormalWisharts)
from mimo_amd.mixtures import BayesianMixtureOfLinearGaussians


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=20000)
    ap.add_argument("--experts", type=int, default=16,
                    help="truncation level (the reference's evaluate_sine.py defaults to 50: --experts 50)")
    ap.add_argument("--iters", type=int, default=150)
    args = ap.parse_args()
    npr.seed(1337)
    N, K, dx, dy = args.rows, args.experts, 1, 1
    x = npr.uniform(-6., 6., size=(N, dx))
    y = 3. * np.sin(x) + 0.2 * (1. + 0.3 * np.abs(x)) * npr.standard_normal((N, dy))

    gating = CategoricalWithStickBreaking(K, TruncatedStickBreaking(K, np.ones(K), 10. * np.ones(K)))
    basis_prior = StackedNormalWisharts(K, dx, np.zeros((K, dx)), 1e-2 * np.ones(K), np.stack(K * [np.eye(dx)]),
                                        (dx + 1.) * np.ones(K) + 1e-8)
    models_prior = StackedMatrixNormalWisharts(K, dx + 1, dy, np.zeros((K, dy, dx + 1)), np.stack(K * [1e-2 * np.eye(dx + 1)]),
                                               np.stack(K * [np.eye(dy)]), (dy + 1.) * np.ones(K) + 1e-8)
    model = BayesianMixtureOfLinearGaussians(K, dx, dy, gating,
                                             StackedGaussiansWithNormalWisharts(K, dx, basis_prior),
                                             StackedLinearGaussiansWithMatrixNormalWisharts(K, dx + 1, dy, models_prior),
                                             scale=True)
    model.init_transform(x, y)                                             # standardise inputs and outputs
    model.resample(x, y, maxiter=25, progress_bar=False)                   # a short Gibbs run as initialisation
    vlb = model.meanfield_coordinate_descent(x, y, randomize=False, maxiter=args.iters, tol=1e-6, progress_bar=False)
    print(f"ELBO: {vlb[0]:.2f} -> {vlb[-1]:.2f} in {len(vlb)} iterations")
    eng = model.engine
    if hasattr(eng, "plan"):          # which kernel family ran the passes (50 experts, the reference's default: the narrow kernels)
        print(f"kernels: softmax pass '{eng.plan(K)['kind']}', label pass '{eng.plan(K, gibbs=True)['kind']}'")

    grid = np.linspace(-6., 6., 25)[:, None]
    mu, var, std = model.meanfield_prediction(grid)
    print("   x     sin     mean    std")
    for g, m, s in zip(grid[:, 0], mu[:, 0], std[:, 0]):
        print(f"{g:6.2f} {3. * np.sin(g):7.3f} {m:7.3f} {s:6.3f}")
    print("RMSE against the noiseless curve: %.4f" % float(np.sqrt(np.mean((mu[:, 0] - 3. * np.sin(grid[:, 0])) ** 2))))


if __name__ == "__main__":
    main()
