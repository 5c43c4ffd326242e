"""Host-side statistics helpers with the reference's names (mimo/utils/stats.py)."""
import numpy as np
import numpy.random as npr


def sample_discrete_from_log(p_log, return_lognorms=False, axis=0, dtype=np.int32, engine=None, u=None,
                             seed=None, sweep=0):
    """mimo/utils/stats.py:8-21 — one categorical draw per slice of `p_log` along `axis`:
    sample = #{k : u cum[-1] > cum[k]} with cum = cumsum(exp(p_log - logsumexp)); optionally the log-normalisers.

    The table goes to the GPU once and the draw runs there (mimo_sample_from_log): one thread per column, the
    cumulative sums never materialised.  Uniforms: by default ONE numpy.random.random call of the reference's shape
    (size 1 along `axis`), so a seeded script draws what the reference draws; `u` supplies them explicitly; `seed`
    (an int) selects the library's counter-based Philox stream instead (counter = (column, sweep)).
    The mixture drivers of this package do not call this function: their label step is fused with the log-density
    evaluation (HipEngine.gibbs_labels) and never builds the (K, N) table."""
    from mimo_amd import engine as _engine
    p_log = np.asarray(p_log, dtype=float)
    if p_log.ndim == 0:
        raise ValueError("p_log must have at least one axis")
    axis = axis % p_log.ndim
    moved = np.moveaxis(p_log, axis, 0)
    rest = moved.shape[1:]
    table = np.ascontiguousarray(moved.reshape(moved.shape[0], -1))
    if u is None and seed is None:
        size = list(p_log.shape)
        size[axis] = 1
        u = npr.random(size=size)            # the reference's call (stats.py:14)
    if u is not None:
        u = np.asarray(u, dtype=float).reshape(-1)
    eng = engine if engine is not None else _engine.default_engine()
    out = eng.sample_from_log(table, u=u, seed=0 if seed is None else seed, sweep=sweep, return_lognorms=return_lognorms)
    labels, lognorms = out if return_lognorms else (out, None)
    labels = labels.reshape(rest).astype(dtype, copy=False)
    return (labels, lognorms.reshape(rest)) if return_lognorms else labels
