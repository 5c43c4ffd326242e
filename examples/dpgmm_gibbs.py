"""Dirichlet-process mixture (truncated stick-breaking) fitted by Gibbs sampling with the labels drawn inside the
kernel (Philox counter = global row, sweep) — the shape of BASELINE.json configs[2] at a size of your choice."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mimo_amd.distributions import (TruncatedStickBreaking, CategoricalWithStickBreaking, StackedNormalWisharts,
                                    StackedGaussiansWithNormalWisharts)
from mimo_amd.mixtures import BayesianMixtureOfGaussians


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=8)
    ap.add_argument("--kmax", type=int, default=256)
    ap.add_argument("--clusters", type=int, default=12)
    ap.add_argument("--sweeps", type=int, default=60)
    args = ap.parse_args()
    rng = np.random.default_rng(7)
    N, D, K = args.rows, args.dim, args.kmax
    centres = rng.normal(0., 6., size=(args.clusters, D))
    X = centres[rng.integers(args.clusters, size=N)] + rng.standard_normal((N, D))

    np.random.seed(1)
    gating = CategoricalWithStickBreaking(K, TruncatedStickBreaking(K, np.ones(K), 5. * np.ones(K)))
    prior = StackedNormalWisharts(K, D, np.zeros((K, D)), 1e-2 * np.ones(K), np.stack(K * [np.eye(D)]), (D + 1.) * np.ones(K) + 1e-8)
    model = BayesianMixtureOfGaussians(gating, StackedGaussiansWithNormalWisharts(K, D, prior))

    t0 = time.perf_counter()
    model.resample(X, init_labels='prior', maxiter=args.sweeps, progress_bar=False, label_rng='philox', seed=2024,
                   param_rng=np.random.Generator(np.random.Philox(99)))
    dt = time.perf_counter() - t0
    counts = np.bincount(model.labels_, minlength=K)
    used = np.flatnonzero(counts > 0.002 * N)
    print(f"{args.sweeps} sweeps over {N} x {D} rows, Kmax = {K}: {dt:.2f} s incl. upload "
          f"({N * K * args.sweeps / dt:.3g} datapoint-component evaluations/s)")
    print(f"components holding more than 0.2 % of the data: {len(used)} (true clusters: {args.clusters})")
    found = model.components.likelihood.mus[used]
    err = np.min(np.linalg.norm(found[:, None, :] - centres[None, :, :], axis=2), axis=1)
    print("distance of their means to the nearest true centre: max %.3f" % err.max())


if __name__ == "__main__":
    main()
