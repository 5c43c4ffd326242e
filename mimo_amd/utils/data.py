"""Host-side data helpers with the reference's names (mimo/utils/data.py)."""
import random
from itertools import islice

import numpy as np


def batches(batch_size, data_size):
    """mimo/utils/data.py:9-12 — yields exactly ONE minibatch of `batch_size` distinct indices
    (SURVEY.md Appendix B #6); same `random.sample` call, so `random.seed` reproduces it."""
    idx_all = random.sample(range(data_size), batch_size)
    idx_iter = iter(idx_all)
    yield from iter(lambda: list(islice(idx_iter, batch_size)), [])


def one_hot(z, K):
    """mimo/utils/data.py:160-169 — dense (K, N) table.  Provided for API parity on small inputs;
    the mixture drivers never build it: hard labels go to HipEngine.label_stats instead."""
    z = np.atleast_1d(z).astype(int)
    assert np.all(z >= 0) and np.all(z < K)
    N, shp = z.size, z.shape
    zoh = np.zeros((K, N))
    zoh[np.ravel(z), np.arange(N)] = 1
    return np.reshape(zoh, (K,) + shp)


def islist(*args):
    return all(isinstance(_arg, list) for _arg in args)
