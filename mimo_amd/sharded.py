"""Data-parallel sharding of the hot path: one process per GPU, N rows split across ranks.

The path shards trivially over rows: every per-datum quantity is local, and the
only exchange per sweep is ONE all-reduce(sum, float64) of the packed statistic block
[K x (1 + Dz + Dz^2)] + 3 ELBO scalars (0.14 MB at K=64, Dz=16) — latency-bound on xGMI, issued as a
single fused call through torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in CPU tests).
After it every rank holds the global statistics and performs the identical O(K D^3) host update, so
no broadcast is needed.

Two routes for that all-reduce on GPUs: torch.distributed's all_reduce on a stream this object owns (default, and
the only one for the CPU test double over gloo); or, with MIMO_SHARDED_NATIVE=1, the library's OWN communicator
(mimo_comm_init: RCCL opened by libmimo_hip.so — inside a PyTorch process the very copy PyTorch loaded — the
collective is enqueued by the C library on the stream of its kernels, right behind them: no second stream, no
cross-stream events, no Python in the path; the unique id travels over torch.distributed once at set-up).  Measured
with one rank at C2 (40 steps, alternating): 6.90 ms per step without a collective, 6.95 / 6.96 ms through torch,
6.93 / 6.99 ms native — the route is not what bounds the step, so the default is the one every PyTorch-ROCm
installation exercises; the native one exists for hosts without PyTorch (include/mimo_hip.h).  Labels / responsibilities stay on the owning rank; the Philox counter uses
the global row index, so labels do not depend on the number of ranks.

The sum over the ranks is taken in RANK ORDER by default: the blocks are all-gathered and added
block 0 first on every rank, so the association — and with it every bit of a sweep — does not depend on the collective
library's choice of algorithm or channel count (MIMO_SHARDED_RANK_ORDER=0: a plain all_reduce(sum)).

Gibbs sweeps draw the K parameter blocks on EVERY rank from the host generator; the ranks only stay consistent if those
streams are identical.  `assert_replicated` (called by the drivers on the first sweep of a run) compares a checksum of the
drawn canonical parameters across the ranks and raises instead of letting a mis-seeded rank diverge silently.

ShardedEngine has the HipEngine interface, so the mixture drivers run unchanged on top of it.
"""
import os

import numpy as np

from mimo_amd.engine import SuffStats


class ShardedEngine:

    def __init__(self, inner, group=None, row_offset=0):
        import torch.distributed as dist
        self._dist = dist
        self.inner = inner
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = getattr(inner, "device", 0)
        self._row0 = int(row_offset)
        self._buf = None
        self._structure = 'full'
        self._xx_global = None
        self._nccl = dist.get_backend(group) == "nccl"
        self._native = False
        self._rank_order = os.environ.get("MIMO_SHARDED_RANK_ORDER", "1") != "0"
        self._checks_left = 2
        if self._nccl and hasattr(inner, "comm_init") and os.environ.get("MIMO_SHARDED_NATIVE", "0") == "1":
            try:
                self._attach_native(inner)
            except Exception as exc:          # librccl not loadable, communicator refused: the torch route still works
                import warnings
                warnings.warn(f"native RCCL communicator unavailable ({exc}); using torch.distributed")
                self._native = False
        self._device_path = hasattr(inner, "estep_device") and self._nccl and not self._native
        if self._device_path:
            # kernel -> all-reduce -> copy to pinned memory are ordered by ONE stream of this object's own: torch's
            # current stream may be the null stream, whose handle (0) means "the context's private stream" to
            # mimo_set_stream — the collective would then not wait for the kernels
            import torch
            self._stream = torch.cuda.Stream(device=self.device)
            inner.set_stream(self._stream.cuda_stream)

    def _attach_native(self, engine):
        """rank 0 creates the RCCL unique id, torch.distributed carries the 128 bytes to the other ranks, every rank
        attaches its context: from here on every statistics call of `engine` returns sums over the ranks."""
        box = [type(engine).comm_unique_id() if self.rank == 0 else None]
        self._dist.broadcast_object_list(box, src=self._dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                                         group=self.group)
        engine.comm_init(box[0], self.rank, self.world)
        self._native = True

    def _nan_share(self, S):
        """The NaN rows' share of the gating counts is a host-side sum over THIS rank's rows: when any rank holds such rows,
        every rank joins one more (K-sized) sum — with or without NaN rows of its own."""
        if S is None or not self._any_nan():       # (one cached scalar all-reduce per data set)
            return S
        local = np.zeros_like(S.n) if getattr(S, 'n_rows', None) is None else S.n_rows - S.n
        S.n_rows = S.n + self._allreduce_array(local)
        return S

    def assert_replicated(self, *arrays, what="parameters"):
        """Raise if `arrays` (host arrays every rank is supposed to hold identically: the parameter blocks a Gibbs sweep drew
        from the host generator) differ between the ranks.  Two 2-element reductions (max and min of a checksum)."""
        import torch
        h = 0.0
        for i, a in enumerate(arrays):
            a = np.ascontiguousarray(a, dtype=np.float64).ravel()
            h += float(np.dot(a, np.cos(np.arange(a.size) + i)))      # (order-sensitive, cheap; NaN propagates)
        t = torch.tensor([h, -h], dtype=torch.float64)
        if self._nccl:
            t = t.to(f"cuda:{self.device}")
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX, group=self.group)
        hi, neg_lo = (float(v) for v in t.cpu())
        if not (hi == -neg_lo):
            raise RuntimeError(
                f"rank {self.rank}: the {what} differ between the ranks (checksums span [{-neg_lo!r}, {hi!r}]). A sharded Gibbs "
                "sweep draws the component blocks on every rank from the host generator: seed numpy.random identically on "
                "all ranks, or pass the same seeded `param_rng` everywhere.")

    def check_replicated_once(self, *arrays, what="parameters"):
        """`assert_replicated` on the first TWO calls after an upload (the drivers call it every sweep): a run that starts from
        `init_labels='posterior'` makes its first label pass with the parameters it was given — identical on every rank — and
        only the second one with blocks drawn from the ranks' host generators."""
        if self._checks_left > 0:
            self._checks_left -= 1
            self.assert_replicated(*arrays, what=what)

    def _any_nan(self):
        if getattr(self, '_any_nan_cache', None) is None:
            self._any_nan_cache = bool(self._allreduce_array(np.array([float(getattr(self.inner, 'n_bad', 0))]))[0] > 0)
        return self._any_nan_cache

    # ---- pass-throughs ---------------------------------------------------------------------------
    @property
    def N(self):
        return self.inner.N

    @property
    def D(self):
        return self.inner.D

    @property
    def n_bad(self):
        """Rows of THIS rank's block that hold a NaN (the host-side fixes of the linear-Gaussian mixtures are row-local:
        mixtures/ilr.py nan_rows_table, lingauss.log_likelihood)."""
        return getattr(self.inner, 'n_bad', 0)

    def nan_rows(self):
        return self.inner.nan_rows()

    def plan(self, K, gibbs=False):
        return self.inner.plan(K, gibbs)

    def tune(self, key, value):
        return self.inner.tune(key, value)

    def predict_device(self, *args, **kwargs):
        return self.inner.predict_device(*args, **kwargs)      # row-local, like predict

    def upload(self, Z_local):
        """Bind this rank's row block."""
        self.inner.upload(Z_local)
        self.inner.set_row_offset(self._row0)
        self._xx_global = None
        self._any_nan_cache = None
        self._checks_left = 2

    def set_row_offset(self, row0):
        self._row0 = int(row0)
        self.inner.set_row_offset(self._row0)

    def spawn(self):
        """A second sharded engine over the same ranks (the SVI minibatch engine): its statistics are all-reduced
        like the main one's, so a minibatch is the union of the ranks' local draws."""
        return ShardedEngine(self.inner.spawn(), self.group, self._row0)      # (attaches its own communicator)

    def global_rows(self, n_local):
        """sum over the ranks of `n_local` (cached per value: one scalar all-reduce)."""
        cache = self.__dict__.setdefault('_rows_cache', {})
        if n_local not in cache:
            cache[n_local] = int(round(float(self._allreduce_array(np.array([float(n_local)]))[0])))
        return cache[n_local]

    def predict(self, *args, **kwargs):
        """Posterior-predictive moments are row-local: every rank predicts its own rows, nothing is exchanged."""
        return self.inner.predict(*args, **kwargs)

    def get_resp(self, K=None):
        return self.inner.get_resp(K)

    def get_logp(self, K=None):
        return self.inner.get_logp(K)

    def get_lse(self):
        return self.inner.get_lse()

    def get_labels(self):
        return self.inner.get_labels()

    # ---- the exchange step -------------------------------------------------------------------------
    def _sum_ranks(self, t):
        """Sum of tensor `t` over the ranks, in place.  Rank order (default): all-gather, then block 0 + block 1 + ... on
        every rank — an association that does not depend on the transport."""
        if not self._rank_order or self.world == 1:
            self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
            return t
        import torch
        parts = [torch.empty_like(t) for _ in range(self.world)]
        self._dist.all_gather(parts, t, group=self.group)
        t.copy_(parts[0])
        for r in range(1, self.world):
            t.add_(parts[r])
        return t

    def _allreduce_array(self, arr):
        """Sum a host float64 array over the ranks (through a device tensor when the backend is RCCL,
        which only reduces device memory)."""
        import torch
        t = torch.from_numpy(np.array(arr, dtype=np.float64))
        if self._nccl:
            return self._sum_ranks(t.to(f"cuda:{self.device}")).cpu().numpy()
        return self._sum_ranks(t).numpy()

    def _allreduce_host(self, S, extra):
        """One sum over the ranks of the packed block + `extra` (+ the K shares of the rows with NaN where any rank has such
        rows: the gating update counts them, engine.SuffStats.n_rows)."""
        K, D = S.sx.shape
        nan = self._any_nan()
        share = [np.zeros(K) if getattr(S, 'n_rows', None) is None else S.n_rows - S.n] if nan else []
        extra = np.asarray(extra, dtype=float)
        if S.sxx is None:      # 'linear' structure: n, sum r z and the pooled second moment of the local rows
            m = K * (1 + D)
            out = self._allreduce_array(np.concatenate([S.n, S.sx.ravel(), np.asarray(S.sxx_total, dtype=float).ravel(), extra] + share))
            G = SuffStats(out[:K].copy(), out[K:m].reshape(K, D).copy(), None, out[m:m + D * D].reshape(D, D).copy())
            rest = out[m + D * D:]
        else:
            out = self._allreduce_array(np.concatenate([S.packed().ravel(), extra] + share))
            G = SuffStats.from_packed(out[:K * (1 + D + D * D)], K, D)
            rest = out[K * (1 + D + D * D):]
        if nan:
            G.n_rows = G.n + rest[len(extra):]
            rest = rest[:len(extra)]
        return G, rest

    # ---- 'linear' structure (one precision for all components) on the device path ---------------------------
    def _linear(self):
        return self._structure == 'linear'

    def _tied(self, W):
        W = np.asarray(W)
        return np.array_equal(W, np.broadcast_to(W[:1], W.shape))

    def _global_xx(self):
        """sum_n z_n z_n' over the rows of ALL ranks: one extra all-reduce of Dz^2 numbers per data set."""
        if self._xx_global is None:
            self._xx_global = self._allreduce_array(self.inner._xx_total())
        return self._xx_global

    def _linear_finish(self, S, sc, W0):
        """What HipEngine adds on one GPU (engine.py, _linear_stats / _linear_scalars), with the pooled moment of
        all shards: the packed block the kernels reduced carries zeros in its second-moment columns."""
        xx = self._global_xx()
        S.sxx, S.sxx_total = None, xx
        if sc is not None:
            corr = - 0.5 * float(np.sum(W0 * xx))
            sc[0] += corr
            sc[1] += corr
        return S, sc

    def _device_buffer(self, K):
        import torch
        D = self.inner.D
        n = K * (1 + D + D * D) + 4
        if self._buf is None or self._buf.numel() != n:
            self._stream.synchronize()             # the previous buffer may still be in flight
            with torch.cuda.stream(self._stream):
                self._buf = torch.zeros(n, dtype=torch.float64, device=f"cuda:{self.device}")
            self._host = torch.empty(n, dtype=torch.float64).pin_memory()
        return self._buf, n - 4

    def _reduce_to_host(self, buf, wait=True):
        """Sum the device block over the ranks and bring it to pinned host memory, behind the kernels on this object's stream."""
        import torch
        with torch.cuda.stream(self._stream):
            self._sum_ranks(buf)
            self._host.copy_(buf, non_blocking=True)
        if wait:
            self._stream.synchronize()

    def set_structure(self, structure):
        """Forwarded to every shard.  Under 'linear' the pooled second moment and the constant the shared quadratic
        term adds to the bound are sums over rows like everything else: the host route all-reduces what the inner
        engine returns for its rows, the device route adds them after the all-reduce (`_linear_finish`)."""
        self._structure = structure
        if hasattr(self.inner, 'set_structure'):
            self.inner.set_structure(structure)        # (native route: the inner engine's pooled moment is already global)

    def estep(self, c, b, W, stats=True, keep_resp=False, keep_logp=False, keep_lse=False, entropy_split=False,
              row_weights=None):
        """`row_weights` are the weights of THIS rank's rows (hierarchical drivers)."""
        if self._native:        # the library already summed the block and the scalars over the ranks
            S, sc = self.inner.estep(c, b, W, stats=stats, keep_resp=keep_resp, keep_logp=keep_logp, keep_lse=keep_lse,
                                     entropy_split=entropy_split, row_weights=row_weights)
            return self._nan_share(S), sc
        if row_weights is not None:
            S, sc = self.inner.estep(c, b, W, stats=stats, keep_resp=keep_resp, keep_logp=keep_logp, keep_lse=keep_lse,
                                     entropy_split=entropy_split, row_weights=row_weights)
            return self._allreduce_host(S, sc) if stats else (None, self._allreduce_scalars(sc))
        linear = self._linear()
        if self._device_path and stats and not (keep_resp or keep_logp or keep_lse or entropy_split) \
                and (not linear or self._tied(W)) and not self._any_nan():
            K = np.asarray(c).shape[0]
            buf, slen = self._device_buffer(K)
            if linear:
                self._global_xx()       # (a first call runs its own pass and all-reduce)
            self.inner.estep_device(c, b, W, buf.data_ptr(), buf.data_ptr() + 8 * slen)
            self._reduce_to_host(buf)
            out = self._host.numpy()
            S, sc = SuffStats.from_packed(out[:slen], K, self.inner.D), out[slen:slen + 3].copy()
            return self._linear_finish(S, sc, np.asarray(W, dtype=float)[0]) if linear else (S, sc)
        S, sc = self.inner.estep(c, b, W, stats=stats, keep_resp=keep_resp, keep_logp=keep_logp, keep_lse=keep_lse,
                                 entropy_split=entropy_split)
        if not stats:
            t = self._allreduce_scalars(sc)
            return None, t
        return self._allreduce_host(S, sc)

    def estep_async(self, c, b, W):
        """Enqueue the fused pass, the RCCL all-reduce of the statistic block and its copy to pinned host
        memory on the shared stream; estep_wait() synchronises.  Host work in between overlaps all three."""
        if self._native:
            self.inner.estep_async(c, b, W)
            self._pending, self._pending_sync = 'native', False
            return
        if not self._device_path or (self._linear() and not self._tied(W)) or self._any_nan():
            self._pending = self.estep(c, b, W)
            self._pending_sync = True
            return
        import torch
        K = np.asarray(c).shape[0]
        buf, slen = self._device_buffer(K)
        W0 = None
        if self._linear():
            self._global_xx()
            W0 = np.array(np.asarray(W, dtype=float)[0])
        self.inner.estep_device(c, b, W, buf.data_ptr(), buf.data_ptr() + 8 * slen)
        self._reduce_to_host(buf, wait=False)
        self._pending = (K, slen, self._stream, W0)
        self._pending_sync = False

    def estep_wait(self):
        p, self._pending = self._pending, None
        if p == 'native':
            S, sc = self.inner.estep_wait()
            return self._nan_share(S), sc
        if self._pending_sync:
            return p
        K, slen, stream, W0 = p
        stream.synchronize()
        out = self._host.numpy()
        S, sc = SuffStats.from_packed(out[:slen], K, self.inner.D), out[slen:slen + 3].copy()
        return (S, sc) if W0 is None else self._linear_finish(S, sc, W0)

    def _allreduce_scalars(self, sc):
        return self._allreduce_array(np.array(sc, dtype=float))

    def gibbs_labels(self, c, b, W, seed=0, sweep=0, u=None, stats=True, return_labels=True, keep_logp=False):
        if self._native:
            labels, S = self.inner.gibbs_labels(c, b, W, seed=seed, sweep=sweep, u=u, stats=stats,
                                                return_labels=return_labels, keep_logp=keep_logp)
            return labels, self._nan_share(S)
        linear = self._linear()
        if self._device_path and stats and u is None and not return_labels and not keep_logp \
                and (not linear or self._tied(W)) and not self._any_nan():
            K = np.asarray(c).shape[0]
            buf, slen = self._device_buffer(K)
            if linear:
                self._global_xx()
            self.inner.gibbs_labels_device(c, b, W, seed, sweep, buf.data_ptr())
            self._reduce_to_host(buf)
            S = SuffStats.from_packed(self._host.numpy()[:slen], K, self.inner.D)
            return None, (self._linear_finish(S, None, None)[0] if linear else S)
        labels, S = self.inner.gibbs_labels(c, b, W, seed=seed, sweep=sweep, u=u, stats=stats,
                                            return_labels=return_labels, keep_logp=keep_logp)
        if S is not None:
            S, _ = self._allreduce_host(S, [])
        return labels, S

    def weighted_stats(self, resp=None, K=None):
        if self._native:
            return self._nan_share(self.inner.weighted_stats(resp, K))
        S, _ = self._allreduce_host(self.inner.weighted_stats(resp, K), [])
        return S

    def label_stats(self, labels, K):
        if self._native:
            return self._nan_share(self.inner.label_stats(labels, K))
        S, _ = self._allreduce_host(self.inner.label_stats(labels, K), [])
        return S

    def random_resp_stats(self, K, seed=0):
        """Random initial responsibilities on every shard (Philox counters use the global row: the draw does not
        depend on the number of ranks), statistics summed over the ranks."""
        if self._native:
            return self._nan_share(self.inner.random_resp_stats(K, seed))
        S, _ = self._allreduce_host(self.inner.random_resp_stats(K, seed), [])
        return S

    def sample_from_log(self, *args, **kwargs):
        return self.inner.sample_from_log(*args, **kwargs)      # row-local

    def table_entropy(self, table=None):
        return float(self._allreduce_scalars([self.inner.table_entropy(table)])[0])


def shard_rows(N, rank, world):
    """Contiguous row block [lo, hi) of rank `rank` (sizes differ by at most one row)."""
    base, rem = divmod(N, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
