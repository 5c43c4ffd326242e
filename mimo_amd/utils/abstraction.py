"""Tuple algebra of sufficient statistics / natural parameters.

Mirrors the behaviour of the reference's `Statistics` (mimo/utils/abstraction.py:7-24):
`prior.nat_param + stats`, `stats * scalar`, element-wise over the tuple (and over lists of arrays
inside the tuple).
"""


def _is_list(*xs):
    return all(isinstance(x, list) for x in xs)


class Statistics(tuple):

    def __new__(cls, x):
        return tuple.__new__(Statistics, x)

    def _zip(self, other, op):
        out = []
        for a, b in zip(self, other):
            out.append([op(u, v) for u, v in zip(a, b)] if _is_list(a, b) else op(a, b))
        return Statistics(out)

    def __add__(self, other):
        return self._zip(other, lambda u, v: u + v)

    def __sub__(self, other):
        return self._zip(other, lambda u, v: u - v)

    def __mul__(self, a):
        return Statistics(a * e for e in self)

    def __rmul__(self, a):
        return Statistics(a * e for e in self)
