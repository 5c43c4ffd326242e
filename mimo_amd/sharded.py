"""Data-parallel sharding of the hot path: one process per GPU, N rows split across ranks.

The path shards trivially over rows: every per-datum quantity is local, and the
only exchange per sweep is ONE all-reduce(sum, float64) of the packed statistic block
[K x (1 + Dz + Dz^2)] + 3 ELBO scalars (0.14 MB at K=64, Dz=16) — latency-bound on xGMI, issued as a
single fused call through torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in CPU tests).
After it every rank holds the global statistics and performs the identical O(K D^3) host update, so
no broadcast is needed.  Labels / responsibilities stay on the owning rank; the Philox counter uses
the global row index, so labels do not depend on the number of ranks.

ShardedEngine has the HipEngine interface, so the mixture drivers run unchanged on top of it.
"""
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
        self._nccl = dist.get_backend(group) == "nccl"
        self._device_path = hasattr(inner, "estep_device") and self._nccl
        if self._device_path:
            import torch
            inner.set_stream(torch.cuda.current_stream().cuda_stream)

    # ---- pass-throughs ---------------------------------------------------------------------------
    @property
    def N(self):
        return self.inner.N

    @property
    def D(self):
        return self.inner.D

    def upload(self, Z_local):
        """Bind this rank's row block."""
        self.inner.upload(Z_local)
        self.inner.set_row_offset(self._row0)

    def set_row_offset(self, row0):
        self._row0 = int(row0)
        self.inner.set_row_offset(self._row0)

    def spawn(self):
        return ShardedEngine(self.inner.spawn() if hasattr(self.inner, "spawn") else type(self.inner)(self.device),
                             self.group, self._row0)

    def get_resp(self, K=None):
        return self.inner.get_resp(K)

    def get_logp(self, K=None):
        return self.inner.get_logp(K)

    def get_lse(self):
        return self.inner.get_lse()

    def get_labels(self):
        return self.inner.get_labels()

    # ---- the exchange step -------------------------------------------------------------------------
    def _allreduce_array(self, arr):
        """Sum a host float64 array over the ranks (through a device tensor when the backend is RCCL,
        which only reduces device memory)."""
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64))
        if self._nccl:
            d = t.to(f"cuda:{self.device}")
            self._dist.all_reduce(d, op=self._dist.ReduceOp.SUM, group=self.group)
            return d.cpu().numpy()
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        return t.numpy()

    def _allreduce_host(self, S, extra):
        K, D = S.sx.shape
        out = self._allreduce_array(np.concatenate([S.packed().ravel(), np.asarray(extra, dtype=float)]))
        return SuffStats.from_packed(out[:K * (1 + D + D * D)], K, D), out[K * (1 + D + D * D):]

    def _device_buffer(self, K):
        import torch
        D = self.inner.D
        n = K * (1 + D + D * D) + 4
        if self._buf is None or self._buf.numel() != n:
            self._buf = torch.zeros(n, dtype=torch.float64, device=f"cuda:{self.device}")
            self._host = torch.empty(n, dtype=torch.float64).pin_memory()
        return self._buf, n - 4

    def set_structure(self, structure):
        """'linear' needs the pooled second moment of ALL shards: the sharded path keeps such blocks on the
        full feature map (same results; the tied fast path is single-GPU for now)."""
        if hasattr(self.inner, 'set_structure'):
            self.inner.set_structure('full' if structure == 'linear' else structure)

    def estep(self, c, b, W, stats=True, keep_resp=False, keep_logp=False, keep_lse=False, entropy_split=False,
              row_weights=None):
        """`row_weights` are the weights of THIS rank's rows (hierarchical drivers)."""
        if row_weights is not None:
            S, sc = self.inner.estep(c, b, W, stats=stats, keep_resp=keep_resp, keep_logp=keep_logp, keep_lse=keep_lse,
                                     entropy_split=entropy_split, row_weights=row_weights)
            return self._allreduce_host(S, sc) if stats else (None, self._allreduce_scalars(sc))
        if self._device_path and stats and not (keep_resp or keep_logp or keep_lse or entropy_split):
            K = np.asarray(c).shape[0]
            buf, slen = self._device_buffer(K)
            self.inner.estep_device(c, b, W, buf.data_ptr(), buf.data_ptr() + 8 * slen)
            self._dist.all_reduce(buf, op=self._dist.ReduceOp.SUM, group=self.group)
            self._host.copy_(buf, non_blocking=False)
            out = self._host.numpy()
            return SuffStats.from_packed(out[:slen], K, self.inner.D), out[slen:slen + 3].copy()
        S, sc = self.inner.estep(c, b, W, stats=stats, keep_resp=keep_resp, keep_logp=keep_logp, keep_lse=keep_lse,
                                 entropy_split=entropy_split)
        if not stats:
            t = self._allreduce_scalars(sc)
            return None, t
        return self._allreduce_host(S, sc)

    def estep_async(self, c, b, W):
        """Enqueue the fused pass, the RCCL all-reduce of the statistic block and its copy to pinned host
        memory on the shared stream; estep_wait() synchronises.  Host work in between overlaps all three."""
        if not self._device_path:
            self._pending = self.estep(c, b, W)
            return
        import torch
        K = np.asarray(c).shape[0]
        buf, slen = self._device_buffer(K)
        self.inner.estep_device(c, b, W, buf.data_ptr(), buf.data_ptr() + 8 * slen)
        self._dist.all_reduce(buf, op=self._dist.ReduceOp.SUM, group=self.group)
        self._host.copy_(buf, non_blocking=True)
        self._pending = (K, slen, torch.cuda.current_stream())

    def estep_wait(self):
        p, self._pending = self._pending, None
        if not self._device_path:
            return p
        K, slen, stream = p
        stream.synchronize()
        out = self._host.numpy()
        return SuffStats.from_packed(out[:slen], K, self.inner.D), out[slen:slen + 3].copy()

    def _allreduce_scalars(self, sc):
        return self._allreduce_array(np.array(sc, dtype=float))

    def gibbs_labels(self, c, b, W, seed=0, sweep=0, u=None, stats=True, return_labels=True, keep_logp=False):
        if self._device_path and stats and u is None and not return_labels and not keep_logp:
            K = np.asarray(c).shape[0]
            buf, slen = self._device_buffer(K)
            self.inner.gibbs_labels_device(c, b, W, seed, sweep, buf.data_ptr())
            self._dist.all_reduce(buf, op=self._dist.ReduceOp.SUM, group=self.group)
            self._host.copy_(buf, non_blocking=False)
            return None, SuffStats.from_packed(self._host.numpy()[:slen], K, self.inner.D)
        labels, S = self.inner.gibbs_labels(c, b, W, seed=seed, sweep=sweep, u=u, stats=stats,
                                            return_labels=return_labels, keep_logp=keep_logp)
        if S is not None:
            S, _ = self._allreduce_host(S, [])
        return labels, S

    def weighted_stats(self, resp=None, K=None):
        S, _ = self._allreduce_host(self.inner.weighted_stats(resp, K), [])
        return S

    def label_stats(self, labels, K):
        S, _ = self._allreduce_host(self.inner.label_stats(labels, K), [])
        return S

    def table_entropy(self, table=None):
        return float(self._allreduce_scalars([self.inner.table_entropy(table)])[0])


def shard_rows(N, rank, world):
    """Contiguous row block [lo, hi) of rank `rank` (sizes differ by at most one row)."""
    base, rem = divmod(N, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
