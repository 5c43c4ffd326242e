"""Host-side data helpers with the reference's names (mimo/utils/data.py)."""
import ctypes as C
import math
import random
from itertools import islice

import numpy as np


_mt = None


def _native_lib():
    from mimo_amd.distributions import composite
    return composite._native()


def _maybe_array(idx, as_array):
    return np.asarray(idx, dtype=np.int64) if as_array else idx


def sample_indices(n, k, as_array=False):
    """random.sample(range(n), k): the same list and the same state of Python's global generator afterwards, computed natively
    (mimo_host_py_sample: both branches of CPython's `sample`; 0.13 ms for 4096 of 4e6) or, without the library, on blocks of the raw
    Mersenne-Twister stream with NumPy (0.35 ms) — CPython's loop costs 0.25 us per getrandbits call through three Python frames:
    1 - 2 ms for the 4096 indices of an SVI minibatch, most of an outer iteration.
    CPython's algorithm (Lib/random.py, `sample`, the branch for populations larger than its set-size threshold):
    j = _randbelow(n) = getrandbits(n.bit_length()) redrawn while >= n, redrawn while already selected — i.e. the first k
    distinct values of the accepted stream in order of first appearance; getrandbits(b <= 32) is one 32-bit output >> (32 - b).
    Anything outside that branch (small populations, n >= 2^32, a replaced generator) goes to random.sample itself."""
    inst = getattr(random, '_inst', None)
    bits = int(n).bit_length()
    setsize = 21 + (4 ** math.ceil(math.log(k * 3, 4)) if k > 5 else 0)
    if type(inst) is not random.Random or not 0 < k <= n or bits > 32 or k < 256:      # (the state hand-over costs ~55 us: 130 indices' worth)
        return _maybe_array(random.sample(range(n), k), as_array)
    version, internal, gauss_next = inst.getstate()
    if version != 3 or len(internal) != 625:
        return _maybe_array(random.sample(range(n), k), as_array)
    lib = _native_lib()
    if lib is not None:                                    # both branches of CPython's sample, natively (mimo_host_py_sample)
        key = np.array(internal[:-1], dtype=np.uint32)
        pos, out = C.c_int(int(internal[-1])), np.empty(k, dtype=np.int64)
        if lib.mimo_host_py_sample(key.ctypes.data_as(C.c_void_p), C.byref(pos), int(n), int(k), 1 if n <= setsize else 0,
                                   out.ctypes.data_as(C.c_void_p)) == 0:
            inst.setstate((version, tuple(key.tolist()) + (pos.value,), gauss_next))
            return out if as_array else out.tolist()
    if n <= setsize or k < 64:
        return _maybe_array(random.sample(range(n), k), as_array)
    global _mt
    if _mt is None:
        _mt = np.random.MT19937()                          # (constructing one seeds it from the OS: 0.1 ms)
    bg = _mt
    start = {'bit_generator': 'MT19937', 'state': {'key': np.array(internal[:-1], dtype=np.uint32), 'pos': int(internal[-1])}}
    bg.state = start
    accept = n / float(1 << bits)
    raw = np.empty(0, dtype=np.uint64)
    while True:
        more = int(k / accept * 1.05) + 256 if raw.size == 0 else raw.size
        raw = np.concatenate([raw, bg.random_raw(more)])
        r = raw >> np.uint64(32 - bits)
        pos = np.flatnonzero(r < n)                       # places of the accepted values in the raw stream
        acc = r[pos]
        srt = np.sort(acc)
        dup = srt[1:][srt[1:] == srt[:-1]]                # values drawn more than once (k^2 / 2n of them: a handful)
        if dup.size > 64:                                 # (dense sampling: let the sort find the first appearances)
            first = np.sort(np.unique(acc, return_index=True)[1])
        else:
            keep = np.ones(acc.size, dtype=bool)
            for v in set(dup.tolist()):
                keep[np.flatnonzero(acc == v)[1:]] = False
            first = np.flatnonzero(keep)                  # first appearances, in stream order
        if first.size >= k:
            break
    take = first[:k]
    out = r[pos[take]]
    consumed = int(pos[take[-1]]) + 1
    bg.state = start
    bg.random_raw(consumed)
    st = bg.state['state']
    inst.setstate((version, tuple(st['key'].tolist()) + (int(st['pos']),), gauss_next))
    out = out.astype(np.int64)
    return out if as_array else out.tolist()


def batches(batch_size, data_size):
    """mimo/utils/data.py:9-12 — yields exactly ONE minibatch of `batch_size` distinct indices
    (SURVEY.md Appendix B #6); the same indices and generator state as the reference's `random.sample` call, so
    `random.seed` reproduces it."""
    idx_all = sample_indices(data_size, batch_size)
    idx_iter = iter(idx_all)
    yield from iter(lambda: list(islice(idx_iter, batch_size)), [])


def one_hot(z, K):
    """Dense indicator table of a label array: out[k, ...] = 1 where z[...] == k (the reference's `one_hot`,
    mimo/utils/data.py:160-169, returns the same (K,) + z.shape array).  For API parity on small inputs only: the mixture
    drivers never build it — hard labels go to HipEngine.label_stats."""
    labels = np.atleast_1d(np.asarray(z)).astype(int)
    if labels.size and (labels.min() < 0 or labels.max() >= K):
        raise AssertionError("labels outside [0, K)")
    return (np.arange(K).reshape((K,) + (1,) * labels.ndim) == labels[None, ...]).astype(float)


def islist(*args):
    return all(isinstance(_arg, list) for _arg in args)
