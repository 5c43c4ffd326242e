"""The outer loop shared by the stochastic variational drivers (gmm.py:300-326, ilr.py:245-277 of the reference): per
iteration ONE minibatch natural-gradient step (`batches` yields a single batch) and the bound over
the full data under the posterior the step produced.

Plain form (sharded engines, CPU doubles): the reference's order, every pass synchronous.

Pipelined form (one device, two contexts: `eng` holds the data set, `beng` the minibatch): the bound pass fills the device for
milliseconds, the minibatch pass for microseconds, and the step between them is host algebra (conjugate update, the posterior
draws of sample_likelihood) that can only start once the minibatch statistics exist.  So the minibatch pass of iteration i + 1
— which needs the posterior of iteration i, exactly what the bound pass of iteration i needs — is enqueued AHEAD of that bound
pass; its statistics come back at once, and the host step of iteration i + 1 runs while the bound of iteration i occupies the
device.  numpy.random / random are consumed in the plain form's order (draws of step i, then the indices of batch i + 1)."""
import numpy as np
import numpy.random as npr

from mimo_amd.utils.data import batches, sample_indices


def _start_stats(beng, size, nbatch):
    resp = npr.rand(size, nbatch)
    resp /= np.sum(resp, axis=0)
    return beng.weighted_stats(resp)


def run(eng, beng, nrows, maxiter, batch_size, randomize, size, upload, canonical, step, prior_terms, tick):
    """upload(batch): rows `batch` -> beng;  canonical() -> (c, b, W) of the current posterior;  step(Sb): the
    natural-gradient step from minibatch statistics;  prior_terms() -> float;  tick(): progress.  Returns the bounds."""
    vlb = []
    if maxiter <= 0:
        return vlb
    pipelined = hasattr(eng, "estep_async") and hasattr(beng, "estep_async") and not hasattr(eng, "inner")
    if not pipelined:
        for i in range(maxiter):
            for batch in batches(batch_size, nrows):
                upload(batch)
                Sb = _start_stats(beng, size, len(batch)) if (i == 0 and randomize is True) else beng.estep(*canonical())[0]
                step(Sb)
            _, sc = eng.estep(*canonical(), stats=False)
            vlb.append(prior_terms() + sc[0])
            tick()
        return vlb

    def next_batch():
        batch = sample_indices(nrows, batch_size, as_array=True)      # what batches() yields, as an index array (no list round trip)
        upload(batch)
        return batch

    batch = next_batch()
    step(_start_stats(beng, size, len(batch)) if randomize is True else beng.estep(*canonical())[0])
    pending = None
    for i in range(maxiter):
        canon = canonical()                              # posterior of iteration i
        more = i + 1 < maxiter
        if more:
            next_batch()
            beng.estep_async(*canon)                     # minibatch pass of iteration i + 1: ahead of the bound in the queue
        if pending is not None:
            vlb.append(pending + eng.estep_wait()[1][0])
            tick()
        eng.estep_async(*canon, stats=False)             # the bound of iteration i
        pending = prior_terms()
        if more:
            step(beng.estep_wait()[0])                   # host algebra of iteration i + 1 under the bound pass
    vlb.append(pending + eng.estep_wait()[1][0])
    tick()
    return vlb
