"""HipEngine — the O(N) side of every mixture driver, on one MI355X.

Thin NumPy-facing wrapper over the C ABI of libmimo_hip.so (include/mimo_hip.h).  The
engine evaluates the canonical form  l[k,n] = c_k + b_k.z_n - 1/2 z_n' W_k z_n  (SURVEY.md §8
row A0) for whatever (c, b, W) the host-side distribution classes derive from their point
estimates (Gibbs / EM) or posterior expectations (mean-field VI), and returns the
responsibility-weighted sufficient statistics those classes' conjugate updates consume.

There is no CPU fallback here: constructing an engine without the HIP library or without a
GPU raises.
"""
import ctypes as C

import numpy as np

from . import _lib


class SuffStats:
    """Packed per-component statistics  n_k, sum_n r z, sum_n r z z'  (float64)."""

    __slots__ = ("n", "sx", "sxx", "sxx_total", "n_rows")

    def __init__(self, n, sx, sxx, sxx_total=None, n_rows=None):
        # sxx is None under the 'linear' structure (one precision shared by all components): only
        # sxx_total = sum_k sum_n r_kn z z' (= sum_n w_n z_n z_n') exists, which is all a tied update uses
        # n_rows: sum_n r_kn over ALL rows when the data holds rows with NaN (they are left out of n / sx / sxx, but
        # the reference's gating update counts them: categorical.py:35-46 on the full label / responsibility table)
        self.n, self.sx, self.sxx, self.sxx_total, self.n_rows = n, sx, sxx, sxx_total, n_rows

    @property
    def gating_counts(self):
        """What the gating update consumes: n, plus the share of the rows with NaN when there are any."""
        return self.n if self.n_rows is None else self.n_rows

    @staticmethod
    def from_packed(S, K, D):
        S = np.asarray(S).reshape(K, 1 + D + D * D)
        return SuffStats(S[:, 0].copy(), S[:, 1:1 + D].copy(), S[:, 1 + D:].reshape(K, D, D).copy())

    def packed(self):
        K, D = self.sx.shape
        return np.concatenate([self.n[:, None], self.sx, self.sxx.reshape(K, D * D)], axis=1)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a, _from_buffer=C.c_char.from_buffer, _addressof=C.addressof):
    """address of the array's first element for a void* argument: through the buffer protocol (0.4 us; data_as() costs 7 us,
    five to eight of them per call of the hot path); read-only or empty arrays take ctypes' route."""
    try:
        return _addressof(_from_buffer(a))
    except (TypeError, ValueError, BufferError):
        return a.ctypes.data_as(C.c_void_p)


class _HostSum:
    """Content checksum (`_word_checksum`) of a host array on a helper thread (the library call releases the GIL)."""

    def __init__(self, Z):
        import threading
        self.Z, self.out, self.err = Z, None, None
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _run(self):
        try:
            self.out = _word_checksum(self.Z)
        except Exception as exc:           # pragma: no cover  (reported by result())
            self.err = exc

    def result(self):
        self.thread.join()
        if self.err is not None:
            raise self.err
        return self.out


class BoundDataGuard:
    """Exact staleness check of the resident rows without a hash on the critical path (bind()).  For arrays above 16 MB bind()
    compares a SAMPLE of the bound host array with what it saw at upload time (a fast negative); when the sample matches it
    starts `_HostSum` on a helper thread and returns.  The first data pass of the call runs meanwhile; before its results
    leave the engine (`checked` methods) the host checksum is compared with the checksum of the rows as they were uploaded
    (`data_checksum`: computed on the device inside the upload's NaN scan).  A mismatch — the caller edited a few elements of
    the bound array in place, which the sample did not see — uploads the array again, warns, and repeats the pass: the caller
    never receives numbers computed on stale rows (the reference re-reads its argument on every call, mimo/mixtures/gmm.py:261)."""
    _verify = None

    def _start_verify(self, Z):
        self._verify = _HostSum(Z)

    def _settle(self):
        """True: the resident rows are the bound array's (or nothing is pending).  False: they were stale and have been replaced."""
        v, self._verify = self._verify, None
        if v is None or v.result() == tuple(self.data_checksum()):
            return True
        import warnings
        warnings.warn("the array bound to the engine was edited in place since it was uploaded (a change the sampled fingerprint "
                      "did not see): uploaded again and the pass repeated", RuntimeWarning, stacklevel=3)
        key = getattr(self, "_bound_key", None)
        self.upload(v.Z)
        self._bound_key = key
        return False


def checked(fn):
    """Method decorator: run, then settle a pending verification of the bound rows; if they were stale, run again on the fresh ones."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        out = fn(self, *args, **kwargs)
        if self._verify is not None and not self._settle():
            out = fn(self, *args, **kwargs)
        return out
    return wrapper


class HipEngine(BoundDataGuard):
    def __init__(self, device=0):
        self._lib = _lib.load()
        self._ctx = C.c_void_p()
        rc = self._lib.mimo_create(C.byref(self._ctx), int(device))
        if rc != 0:
            msg = self._lib.mimo_last_error(None).decode()
            self._ctx = None
            raise _lib.MimoHipError(f"mimo_create failed ({rc}): {msg}")
        self.device = int(device)
        self.N = 0
        self.D = 0
        self._keepalive = None

    # -- plumbing -------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            msg = self._lib.mimo_last_error(self._ctx).decode()
            if rc == _lib.E_INVALID:
                raise ValueError(msg)
            raise _lib.MimoHipError(f"libmimo_hip error {rc}: {msg}")

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.mimo_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def spawn(self):
        """A second, independent engine on the same device (own context, stream and resident data): the SVI
        drivers keep the full data set on one and push minibatches through the other."""
        return HipEngine(self.device)

    def global_rows(self, n_local):
        """Rows over all shards for `n_local` rows here (one GPU: itself; ShardedEngine sums over the ranks)."""
        return int(n_local)

    # -- sharding through the library's own RCCL communicator (hosts without torch.distributed) ----------
    @staticmethod
    def comm_unique_id():
        """128-byte id of a new communicator (rank 0 creates it, the application hands it to the other ranks)."""
        lib = _lib.load()
        buf = C.create_string_buffer(128)
        rc = lib.mimo_comm_unique_id(buf)
        if rc != 0:
            raise _lib.MimoHipError(f"mimo_comm_unique_id failed ({rc}): {lib.mimo_last_error(None).decode()}")
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        """Attach this engine to the communicator: every later pass returns statistics summed over the ranks
        (one RCCL all-reduce of the packed block per pass, mimo_comm_init)."""
        self._check(self._lib.mimo_comm_init(self._ctx, bytes(unique_id), int(rank), int(world)))

    def comm_destroy(self):
        self._check(self._lib.mimo_comm_destroy(self._ctx))

    def set_stream(self, stream_ptr):
        self._check(self._lib.mimo_set_stream(self._ctx, C.c_void_p(stream_ptr or 0)))

    def set_row_offset(self, row0):
        self._check(self._lib.mimo_set_row_offset(self._ctx, int(row0)))

    def set_structure(self, structure):
        """'full' (symmetric W), 'diag' (diagonal W: the 2 Dz + 1 feature kernels) or 'linear' (all W_k equal:
        the Dz + 1 feature kernels; this class adds the data constants the shared quadratic term contributes,
        see mimo_set_structure in include/mimo_hip.h)."""
        code = {'full': 0, 'diag': 1, 'linear': 2}[structure]
        if getattr(self, '_structure', 0) != code:
            self._check(self._lib.mimo_set_structure(self._ctx, code))
            self._structure = code

    # -- 'linear' structure: what the shared quadratic term -1/2 z'W z adds back -------------------
    class _AsFull:
        def __init__(self, eng):
            self.eng = eng

        def __enter__(self):
            self.prev = {0: 'full', 1: 'diag', 2: 'linear'}[getattr(self.eng, '_structure', 0)]
            self.eng.set_structure('full')

        def __exit__(self, *a):
            self.eng.set_structure(self.prev)

    def _xx(self, weights=None):
        """sum_n w_n z_n z_n' (D, D): one pass with a single unit-responsibility component, full structure.  For a
        weight vector the result is kept while the vector's content (fingerprint) and the data set stay the same: the
        hierarchical drivers pass the same weights every iteration."""
        key = None
        if weights is not None:
            wv, base = _weights_key(weights)
            key = base + (id(self._keepalive), self.N, getattr(self, '_upload_count', 0))
            hit = getattr(self, '_xxw_cache', None)
            if hit is not None and hit[0] == key:
                return hit[1]
        with HipEngine._AsFull(self):
            w = np.ones((1, self.N)) if weights is None else _f64(weights).reshape(1, -1)
            out = self.weighted_stats(w).sxx[0]
        if key is not None:
            self._xxw_cache = (key, out)
        return out

    def _xx_total(self):
        if getattr(self, '_xx_cache', None) is None:
            self._xx_cache = self._xx()
        return self._xx_cache

    def _linear(self):
        return getattr(self, '_structure', 0) == 2

    def _linear_stats(self, S, total):
        if S is not None:
            S.sxx, S.sxx_total = None, total
        return S

    def _linear_scalars(self, sc, W0):
        corr = - 0.5 * float(np.sum(W0 * self._xx_total()))     # sum_n -1/2 z_n'W z_n = -1/2 tr(W XX)
        sc = np.array(sc, dtype=float)
        sc[0] += corr
        sc[1] += corr          # sum r l gains the same constant (sum_k r = 1); NaN stays NaN
        return sc

    def tune(self, key, value):
        """Launch-geometry override for tests and experiments (mimo_tune): 'num_cu' (0: the device's own), 'sorted_range'."""
        self._check(self._lib.mimo_tune(self._ctx, key.encode(), int(value)))

    def profile(self, enable=True):
        self._check(self._lib.mimo_profile(self._ctx, 1 if enable else 0))

    def profile_read(self, reset=True):
        ms, n = C.c_double(), C.c_int64()
        self._check(self._lib.mimo_profile_read(self._ctx, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value

    def profile_kernels(self):
        """{kernel name: {'ms': total device time, 'launches': n}} since the last profile_read(reset=True)."""
        buf = C.create_string_buffer(2048)
        self._check(self._lib.mimo_profile_kernels(self._ctx, buf, len(buf)))
        out = {}
        for ln in buf.value.decode().splitlines():
            name, ms, n = ln.split("\t")
            out[name] = {"ms": float(ms), "launches": int(n)}
        return out

    def shader_clock_mhz(self):
        """Median shader clock under a short float64 load, from the in-kernel counters (mimo_shader_clock_mhz)."""
        out = C.c_double()
        self._check(self._lib.mimo_shader_clock_mhz(self._ctx, C.byref(out)))
        return out.value

    def plan(self, K, gibbs=False):
        """How a pass with K components runs on the resident data (mimo_plan): kind, kernels, HBM passes."""
        o = (C.c_int64 * 8)()
        self._check(self._lib.mimo_plan(self._ctx, int(K), 1 if gibbs else 0, o))
        kind = {1: "fused", 2: "two-stage", 3: "small", 4: "rowwave", 5: "rowwave-vi", 6: "narrow", 7: "mid"}.get(o[0], "?")
        kernel = {1: "mimo::fused_kernel", 2: "E-step (mimo::wide_estep_kernel / estep_chunked_kernel) + statistics per column group (mimo::wide_stats_kernel / fused_kernel)",
                  3: "mimo::small_kernel", 4: "mimo::gibbs_rowwave_kernel + mimo::label_stats_kernel",
                  5: "mimo::vi_rowwave_kernel",
                  6: "mimo::narrow_kernel (+ mimo::label_stats_kernel for a label pass)",
                  7: "mimo::mid_kernel"}.get(o[0], "?")
        return {"kind": kind, "kernel": kernel, "kernels_per_pass": int(o[1]), "table_in_hbm": bool(o[2]),
                "table_reads": int(o[3]), "data_passes": int(o[4]), "label_passes": int(o[5]),
                "workgroups": int(o[6]), "compute_units": int(o[7])}

    # -- data -----------------------------------------------------------------------------
    def upload(self, Z):
        """Z: (N, Dz) float64 host array (copied once) or a CUDA/HIP torch tensor (borrowed)."""
        self._xx_cache = None
        if hasattr(Z, "data_ptr") and getattr(Z, "is_cuda", False):
            import torch
            if Z.dtype != torch.float64 or not Z.is_contiguous() or Z.dim() != 2:
                raise ValueError("device data must be a contiguous (N, Dz) float64 tensor")
            self._keepalive = Z
            self.N, self.D = int(Z.shape[0]), int(Z.shape[1])
            self._check(self._lib.mimo_attach(self._ctx, C.c_void_p(Z.data_ptr()), self.N, self.D))
            self._after_upload()
            return
        Z = _f64(Z)
        if Z.ndim != 2:
            raise ValueError("data must be (N, Dz)")
        self.N, self.D = int(Z.shape[0]), int(Z.shape[1])
        self._keepalive = None
        self._check(self._lib.mimo_upload(self._ctx, _ptr(Z), self.N, self.D))
        self._after_upload()

    # -- rows with NaN (dropped from the statistics, normaliser-only log-density: mimo_nan_info) -------------
    def data_checksum(self):
        """Content checksum of the rows as upload() received them (mimo_data_checksum; `_word_checksum` of the host array)."""
        out = (C.c_uint64 * 2)()
        self._check(self._lib.mimo_data_checksum(self._ctx, out))
        return (int(out[0]), int(out[1]))

    def _after_upload(self):
        self._verify = None
        self._w_key = None
        self._xxw_cache = None
        self._upload_count = getattr(self, '_upload_count', 0) + 1
        nb = C.c_int64()
        self._check(self._lib.mimo_nan_info(self._ctx, C.byref(nb), None, 0, None))
        self.n_bad, self._bad_rows = int(nb.value), None

    def nan_rows(self):
        """Indices of the resident rows that hold a NaN."""
        if self._bad_rows is None:
            if not getattr(self, 'n_bad', 0):
                self._bad_rows = np.zeros(0, dtype=np.int64)
            else:
                m = np.empty(self.N)
                self._check(self._lib.mimo_nan_info(self._ctx, None, _ptr(m), 0, None))
                self._bad_rows = np.flatnonzero(m == 0.)
        return self._bad_rows

    def _nan_label_counts(self, K):
        out = np.zeros(K, dtype=np.int64)
        self._check(self._lib.mimo_nan_info(self._ctx, None, None, int(K), _ptr(out)))
        return out.astype(float)

    def _nan_softmax_share(self, S, c):
        """softmax pass on data with NaN rows: each of them carries the responsibilities softmax_k(c_k)."""
        if S is not None and getattr(self, 'n_bad', 0):
            c = np.asarray(c, dtype=float)
            m = np.max(c)
            e = np.exp(c - m) if np.isfinite(m) else np.zeros_like(c)
            S.n_rows = S.n + self.n_bad * e / max(np.sum(e), 1e-300)
        return S

    # -- hot path -------------------------------------------------------------------------
    def _params(self, c, b, W):
        c, b, W = _f64(c), _f64(b), _f64(W)
        K = c.shape[0]
        if b.shape != (K, self.D) or W.shape != (K, self.D, self.D):
            raise ValueError(f"parameter shapes {c.shape}, {b.shape}, {W.shape} do not match K={K}, Dz={self.D}")
        return c, b, W, K

    @checked
    def estep(self, c, b, W, stats=True, keep_resp=False, keep_logp=False, keep_lse=False, entropy_split=False,
              row_weights=None):
        """Fused E-step.  Returns (SuffStats | None, scalars[3]); scalars[1:] are NaN unless
        entropy_split (or a keep_* flag) is set.  `row_weights` (N,): the statistics are those of
        r_kn * w_n, tables and scalars stay unweighted (hgmm.py:199-207)."""
        if self._linear():
            Wa = _f64(W)
            if keep_logp or keep_lse or not np.array_equal(Wa, np.broadcast_to(Wa[:1], Wa.shape)):
                with HipEngine._AsFull(self):      # per-datum tables carry the quadratic term itself
                    return self._estep(c, b, W, stats, keep_resp, keep_logp, keep_lse, entropy_split, row_weights)
            S, sc = self._estep(c, b, W, stats, keep_resp, False, False, entropy_split, row_weights)
            total = None
            if stats:
                total = self._xx_total() if row_weights is None else self._xx(row_weights)
            return self._linear_stats(S, total), self._linear_scalars(sc, Wa[0])
        return self._estep(c, b, W, stats, keep_resp, keep_logp, keep_lse, entropy_split, row_weights)

    def _estep(self, c, b, W, stats, keep_resp, keep_logp, keep_lse, entropy_split, row_weights):
        c, b, W, K = self._params(c, b, W)
        flags = ((_lib.F_KEEP_RESP if keep_resp else 0) | (_lib.F_KEEP_LOGP if keep_logp else 0)
                 | (_lib.F_KEEP_LSE if keep_lse else 0) | (0 if stats else _lib.F_NO_STATS)
                 | (_lib.F_ENTROPY_SPLIT if entropy_split else 0))
        S = np.empty((K, 1 + self.D + self.D * self.D)) if stats else None
        sc = np.empty(3)
        if row_weights is not None and stats:
            w, base = _weights_key(row_weights)
            if w.shape[0] != self.N:
                raise ValueError(f"row_weights has {w.shape[0]} entries, data has {self.N} rows")
            # the drivers pass the same weight vector every iteration: it stays on the device while its content does
            # (the drivers freeze it once per call — FrozenWeights —, a plain array is hashed here, every byte)
            wkey = base + (id(self._keepalive), self.N)
            resident = getattr(self, '_w_key', None) == wkey
            rc = self._lib.mimo_estep_weighted(self._ctx, _ptr(c), _ptr(b), _ptr(W), K, _ptr(w),
                                               flags | (_lib.F_WEIGHTS_RESIDENT if resident else 0), _ptr(S), _ptr(sc))
            self._w_key = wkey if rc == 0 else None
            if rc == _lib.E_UNSUPPORTED:      # two-stage shapes take their weights as a table
                self._check(self._lib.mimo_estep(self._ctx, _ptr(c), _ptr(b), _ptr(W), K,
                                                 flags | _lib.F_KEEP_RESP | _lib.F_NO_STATS, None, _ptr(sc)))
                self._K = K
                return self._nan_softmax_share(self.weighted_stats(self.get_resp(K) * w[None, :]), c), sc
            self._check(rc)
        else:
            self._check(self._lib.mimo_estep(self._ctx, _ptr(c), _ptr(b), _ptr(W), K, flags,
                                             _ptr(S) if stats else None, _ptr(sc)))
        self._K = K
        return (self._nan_softmax_share(SuffStats.from_packed(S, K, self.D), c) if stats else None), sc

    def estep_async(self, c, b, W, row_weights=None, stats=True):
        """Enqueue the fused E-step and return immediately; estep_wait() returns (SuffStats, scalars).
        The host can do its own O(K D^3) work (ELBO prior terms) while the data pass runs.  `row_weights` as in estep()
        (two-stage shapes, which take their weights as a table, run synchronously here and hand the result to estep_wait)."""
        c, b, W, K = self._params(c, b, W)
        self._async_args = (c, b, W, row_weights, stats) if self._verify is not None else None
        self._async_stats = bool(stats)
        self._async_W0 = None
        self._async_done = None
        if self._linear():
            if not np.array_equal(W, np.broadcast_to(W[:1], W.shape)):
                self.set_structure('full')      # not a tied block after all: stays full until the caller re-binds
            else:
                (self._xx_total() if row_weights is None else self._xx(row_weights))     # (a first call runs its own pass: before the asynchronous one)
                self._async_W0 = W[0].copy()
        self._async_wts = row_weights
        if row_weights is not None:
            w, base = _weights_key(row_weights)
            if w.shape[0] != self.N:
                raise ValueError(f"row_weights has {w.shape[0]} entries, data has {self.N} rows")
            wkey = base + (id(self._keepalive), self.N)
            resident = getattr(self, '_w_key', None) == wkey
            rc = self._lib.mimo_estep_weighted(self._ctx, _ptr(c), _ptr(b), _ptr(W), K, _ptr(w),
                                               _lib.F_ASYNC | (_lib.F_WEIGHTS_RESIDENT if resident else 0), None, None)
            self._w_key = wkey if rc == 0 else None
            if rc == _lib.E_UNSUPPORTED:
                W0, self._async_W0 = self._async_W0, None
                self._async_done = self.estep(c, b, W, row_weights=row_weights)
                return
            self._check(rc)
        else:
            self._check(self._lib.mimo_estep(self._ctx, _ptr(c), _ptr(b), _ptr(W), K,
                                             _lib.F_ASYNC | (0 if stats else _lib.F_NO_STATS), None, None))
        self._K = K
        self._async_K = K
        self._async_c = np.array(c) if getattr(self, 'n_bad', 0) else None

    def estep_wait(self):
        if getattr(self, '_async_done', None) is not None:
            out, self._async_done = self._async_done, None
            return out
        out = self._estep_wait()
        if self._verify is not None and getattr(self, '_async_args', None) is not None and not self._settle():
            self.estep_async(*self._async_args)          # stale rows: the pass again on the fresh upload
            out = self._estep_wait()
        return out

    def _finish_async(self, S, sc):
        if getattr(self, '_async_c', None) is not None:
            self._nan_softmax_share(S, self._async_c)
        if getattr(self, '_async_W0', None) is not None:
            wts = getattr(self, '_async_wts', None)
            return self._linear_stats(S, self._xx_total() if wts is None else self._xx(wts)), self._linear_scalars(sc, self._async_W0)
        return S, sc

    def _estep_wait(self):
        K = self._async_K
        sc = np.empty(3)
        if not getattr(self, '_async_stats', True):        # scalars only (the full-data bound of the SVI drivers)
            self._check(self._lib.mimo_wait(self._ctx, None, _ptr(sc)))
            if getattr(self, '_async_W0', None) is not None:
                sc = self._linear_scalars(sc, self._async_W0)
            return None, sc
        S = np.empty((K, 1 + self.D + self.D * self.D))
        self._check(self._lib.mimo_wait(self._ctx, _ptr(S), _ptr(sc)))
        return self._finish_async(SuffStats.from_packed(S, K, self.D), sc)

    def estep_device(self, c, b, W, S_dev_ptr, scalars_dev_ptr):
        """Asynchronous fused E-step writing packed S / scalars to device pointers."""
        c, b, W, K = self._params(c, b, W)
        self._check(self._lib.mimo_estep(self._ctx, _ptr(c), _ptr(b), _ptr(W), K, _lib.F_DEVICE_OUT,
                                         C.c_void_p(S_dev_ptr), C.c_void_p(scalars_dev_ptr)))
        self._K = K

    @checked
    def gibbs_labels(self, c, b, W, seed=0, sweep=0, u=None, stats=True, return_labels=True,
                     keep_logp=False):
        """Fused Gibbs label step.  Returns (labels int32 | None, SuffStats | None)."""
        c, b, W, K = self._params(c, b, W)
        if self._linear() and (keep_logp or not np.array_equal(W, np.broadcast_to(W[:1], W.shape))):
            with HipEngine._AsFull(self):
                return self.gibbs_labels(c, b, W, seed, sweep, u, stats, return_labels, keep_logp)
        flags = (0 if stats else _lib.F_NO_STATS) | (_lib.F_KEEP_LOGP if keep_logp else 0)
        S = np.empty((K, 1 + self.D + self.D * self.D)) if stats else None
        labels = np.empty(self.N, dtype=np.int32) if return_labels else None
        if u is not None:
            u = _f64(u).reshape(-1)
            if u.shape[0] != self.N:
                raise ValueError("u must hold one uniform per datum")
            self._w_key = None          # (the uniforms take the device buffer the row weights lived in)
        self._check(self._lib.mimo_gibbs_labels(
            self._ctx, _ptr(c), _ptr(b), _ptr(W), K, int(seed), int(sweep),
            _ptr(u) if u is not None else None, flags,
            _ptr(labels) if return_labels else None, _ptr(S) if stats else None))
        self._K = K
        S = SuffStats.from_packed(S, K, self.D) if stats else None
        if stats and getattr(self, 'n_bad', 0):
            S.n_rows = S.n + self._nan_label_counts(K)
        if self._linear() and stats:      # every row carries exactly one label: the second moments add up to XX
            S = self._linear_stats(S, self._xx_total())
        return labels, S

    def gibbs_labels_device(self, c, b, W, seed, sweep, S_dev_ptr):
        c, b, W, K = self._params(c, b, W)
        self._check(self._lib.mimo_gibbs_labels(
            self._ctx, _ptr(c), _ptr(b), _ptr(W), K, int(seed), int(sweep), None,
            _lib.F_DEVICE_OUT, None, C.c_void_p(S_dev_ptr)))
        self._K = K

    @checked
    def weighted_stats(self, resp=None, K=None):
        """Statistics for arbitrary (K,N) weights; resp=None reuses the resident table."""
        if resp is None:
            K = int(K if K is not None else self._K)
            p = None
        else:
            resp = _f64(resp)
            if resp.ndim != 2 or resp.shape[1] != self.N:
                raise ValueError("weights must be (K, N)")
            K = resp.shape[0]
            p = _ptr(resp)
        S = np.empty((K, 1 + self.D + self.D * self.D))
        self._check(self._lib.mimo_weighted_stats(self._ctx, p, K, 0, _ptr(S)))
        S = SuffStats.from_packed(S, K, self.D)
        if getattr(self, 'n_bad', 0):
            bad = self.nan_rows()
            S.n_rows = S.n + (np.sum(resp[:, bad], axis=1) if resp is not None else np.sum(self.get_resp_columns(bad, K), axis=1))
        if self._linear():     # sum_k r_kn is the weight of row n in the pooled second moment (1 for responsibilities)
            S = self._linear_stats(S, self._xx_total() if resp is None else self._xx(np.sum(resp, axis=0)))
        return S

    @checked
    def label_stats(self, labels, K):
        """Statistics of hard labels (no one-hot table); labels=None reuses the resident draw."""
        K = int(K)
        if labels is None:
            p = None
        else:
            labels = np.ascontiguousarray(labels, dtype=np.int32).reshape(-1)
            if labels.shape[0] != self.N:
                raise ValueError("labels must hold one entry per datum")
            if labels.size and (labels.min() < 0 or labels.max() >= K):
                raise ValueError("labels out of range")  # mirrors the assert in one_hot (data.py:162)
            p = _ptr(labels)
        S = np.empty((K, 1 + self.D + self.D * self.D))
        self._check(self._lib.mimo_label_stats(self._ctx, p, K, 0, _ptr(S)))
        S = SuffStats.from_packed(S, K, self.D)
        if getattr(self, 'n_bad', 0):
            S.n_rows = S.n + self._nan_label_counts(K)
        return self._linear_stats(S, self._xx_total()) if self._linear() else S

    @checked
    def random_resp_stats(self, K, seed=0):
        """Statistics of random initial responsibilities drawn on the device (mimo_random_resp_stats): the
        randomize=True start of the drivers without K N host uniforms; the table stays resident (get_resp)."""
        K = int(K)
        S = np.empty((K, 1 + self.D + self.D * self.D))
        self._check(self._lib.mimo_random_resp_stats(self._ctx, K, int(seed), 0, _ptr(S)))
        self._K = K
        S = SuffStats.from_packed(S, K, self.D)
        if getattr(self, 'n_bad', 0):        # the share of the rows with NaN: their random responsibilities (the table is resident)
            S.n_rows = S.n + np.sum(self.get_resp_columns(self.nan_rows(), K), axis=1)
        return self._linear_stats(S, self._xx_total()) if self._linear() else S

    def sample_from_log(self, logp=None, K=None, u=None, seed=0, sweep=0, return_lognorms=False):
        """Categorical draw per column of a (K, N) log-probability table (mimo_sample_from_log): a host array, or
        None for the log-density table a call with keep_logp=True left on the device.  Returns labels (N,) int32
        [, lognorms (N,)]."""
        if logp is None:
            K, N, p = int(K if K is not None else self._K), self.N, None
        else:
            logp = _f64(logp)
            if logp.ndim != 2:
                raise ValueError("log-probabilities must be (K, N)")
            K, N = logp.shape
            p = _ptr(logp)
        if u is not None:
            u = _f64(u).reshape(-1)
            if u.shape[0] != N:
                raise ValueError("u must hold one uniform per column")
        labels = np.empty(N, dtype=np.int32)
        ln = np.empty(N) if return_lognorms else None
        self._w_key = None
        self._check(self._lib.mimo_sample_from_log(self._ctx, p, K, N, _ptr(u) if u is not None else None, int(seed), int(sweep),
                                                   0, _ptr(labels), _ptr(ln) if ln is not None else None))
        return (labels, ln) if return_lognorms else labels

    def table_entropy(self, table=None):
        """-sum t log t of a (K,N) host table (None: the resident responsibilities)."""
        out = C.c_double()
        if table is None:
            self._check(self._lib.mimo_table_entropy(self._ctx, None, 0, 0, C.byref(out)))
        else:
            table = _f64(table)
            self._check(self._lib.mimo_table_entropy(self._ctx, _ptr(table), table.size, 0, C.byref(out)))
        return out.value

    @staticmethod
    def _host_out(shape):
        """Output array for a large copy-out: page-locked (through PyTorch's caching host allocator) when it is big enough for the
        transfer rate to matter — a device-to-host copy into pageable memory runs at a third of the PCIe rate."""
        n = int(np.prod(shape))
        if n >= (1 << 20):
            try:
                import torch
                return torch.empty(shape, dtype=torch.float64, pin_memory=True).numpy()     # (the array keeps the tensor alive)
            except Exception:
                pass
        return np.empty(shape)

    @checked
    def predict(self, c, b, W, M, Q, Cc, affine=True, mode='average', y=None, P=None, ld=None, variance='full'):
        """Posterior-predictive mixture moments of every resident row (mimo_predict).
        Returns (mu (N,dy), covar (N,dy,dy), nlpd (N) | None); with variance='diagonal' the second entry is the pair
        (var (N,dy), std (N,dy)) computed on the device — what the reference's callers read (ilr.py:411-417) — and a quarter
        (dy = 4) of the bytes cross PCIe."""
        c, b, W, K = self._params(c, b, W)
        M, Q, Cc = _f64(M), _f64(Q), _f64(Cc)
        dy, dc = M.shape[1], self.D + (1 if affine else 0)
        if M.shape != (K, dy, dc) or Q.shape != (K, dc, dc) or Cc.shape != (K, dy, dy):
            raise ValueError(f"predictive blocks {M.shape}, {Q.shape}, {Cc.shape} do not match K={K}, dy={dy}, dc={dc}")
        if mode not in ('average', 'mode'):
            raise NotImplementedError(mode)
        if variance not in ('full', 'diagonal'):
            raise ValueError(variance)
        diag = variance == 'diagonal'
        mu = self._host_out((self.N, dy))
        covar = self._host_out((2, self.N, dy)) if diag else self._host_out((self.N, dy, dy))
        nlpd = None
        if y is not None:
            y, P, ld = _f64(y).reshape(self.N, dy), _f64(P), _f64(ld)
            if P.shape != (K, dy, dy) or ld.shape != (K,):
                raise ValueError("nlpd needs P (K,dy,dy) and ld (K,)")
            nlpd = np.empty(self.N)
        self._check(self._lib.mimo_predict_flags(
            self._ctx, _ptr(c), _ptr(b), _ptr(W), K, _ptr(M), _ptr(Q), _ptr(Cc), dy, 1 if affine else 0,
            0 if mode == 'average' else 1, _ptr(y) if y is not None else None, _ptr(P) if y is not None else None,
            _ptr(ld) if y is not None else None, _ptr(mu), _ptr(covar), _ptr(nlpd) if y is not None else None,
            _lib.F_DIAG_VAR if diag else 0))
        return mu, ((covar[0], covar[1]) if diag else covar), nlpd

    def predict_device(self, c, b, W, M, Q, Cc, mu_ptr, covar_ptr, affine=True, mode='average', y_ptr=None, P=None, ld=None,
                       nlpd_ptr=None):
        """mimo_predict_flags with device pointers for the outputs (and for y): mu (N, dy), covar (N, dy, dy) and nlpd (N) are
        written on the device — nothing but the K parameter blocks crosses PCIe — and the call returns without waiting; the
        results are ordered on the engine's stream (set_stream).  Pointers: e.g. torch tensors' data_ptr()."""
        c, b, W, K = self._params(c, b, W)
        M, Q, Cc = _f64(M), _f64(Q), _f64(Cc)
        dy, dc = M.shape[1], self.D + (1 if affine else 0)
        if M.shape != (K, dy, dc) or Q.shape != (K, dc, dc) or Cc.shape != (K, dy, dy):
            raise ValueError(f"predictive blocks {M.shape}, {Q.shape}, {Cc.shape} do not match K={K}, dy={dy}, dc={dc}")
        if mode not in ('average', 'mode'):
            raise NotImplementedError(mode)
        with_y = y_ptr is not None
        if with_y:
            P, ld = _f64(P), _f64(ld)
            if P.shape != (K, dy, dy) or ld.shape != (K,) or nlpd_ptr is None:
                raise ValueError("nlpd needs y, P (K,dy,dy), ld (K,) and an output pointer")
        self._check(self._lib.mimo_predict_flags(
            self._ctx, _ptr(c), _ptr(b), _ptr(W), K, _ptr(M), _ptr(Q), _ptr(Cc), dy, 1 if affine else 0,
            0 if mode == 'average' else 1, C.c_void_p(y_ptr) if with_y else None, _ptr(P) if with_y else None,
            _ptr(ld) if with_y else None, C.c_void_p(mu_ptr), C.c_void_p(covar_ptr), C.c_void_p(nlpd_ptr) if with_y else None,
            _lib.F_DEVICE_OUT | (_lib.F_DEVICE_IN if with_y else 0)))

    # -- copy-outs ------------------------------------------------------------------------
    def get_resp(self, K=None):
        out = np.empty((int(K if K is not None else self._K), self.N))
        self._check(self._lib.mimo_get_resp(self._ctx, _ptr(out)))
        return out

    def get_resp_columns(self, rows, K=None):
        """(K, len(rows)) block of the resident responsibility table (mimo_get_resp_columns): a few columns, not K N doubles."""
        rows = np.ascontiguousarray(rows, dtype=np.int64).reshape(-1)
        out = np.empty((int(K if K is not None else self._K), rows.shape[0]))
        self._check(self._lib.mimo_get_resp_columns(self._ctx, _ptr(rows), rows.shape[0], _ptr(out)))
        return out

    def get_logp(self, K=None):
        out = np.empty((int(K if K is not None else self._K), self.N))
        self._check(self._lib.mimo_get_logp(self._ctx, _ptr(out)))
        return out

    def get_lse(self):
        out = np.empty(self.N)
        self._check(self._lib.mimo_get_lse(self._ctx, _ptr(out)))
        return out

    def get_labels(self):
        out = np.empty(self.N, dtype=np.int32)
        self._check(self._lib.mimo_get_labels(self._ctx, _ptr(out)))
        return out


def philox_uniforms(seed, rows, sweep):
    """Host mirror of the in-kernel Philox4x32-10 stream (one uniform per global row index)."""
    lib = _lib.load()
    return np.array([lib.mimo_philox_uniform(int(seed), int(r), int(sweep)) for r in rows])


# ---------------------------------------------------------------------------------------------
# data binding used by the reference-shaped array methods (log_likelihood(x), weighted_statistics
# (x, w), ...): the array last bound stays resident, so passing the SAME array again costs nothing.
# ---------------------------------------------------------------------------------------------
_FULL_HASH_BYTES = 1 << 21       # arrays up to 2 MB are fingerprinted in full with a CRC
_FULL_SUM_BYTES = 1 << 24        # ... up to 16 MB in full with a 64-bit word checksum (memory-bound: a few ms)
_SAMPLE_ELEMS = 8192


def _exact_mode():
    import os
    return os.environ.get("MIMO_BIND_EXACT", "0") == "1"


def _verify_mode():
    """Background verification of large bound arrays (BoundDataGuard); MIMO_BIND_VERIFY=0 leaves the sampled fingerprint alone."""
    import os
    return os.environ.get("MIMO_BIND_VERIFY", "1") != "0"


def _word_checksum(Z):
    """Position-dependent checksum over EVERY byte of a C-contiguous array (mimo_host_checksum, include/mimo_hip.h): with w_i the
    8-byte words (the tail zero-extended), m_i = w_i ^ (w_i >> 32) and nw their number, (sum_i m_i, sum_i (nw - i) m_i) mod 2^64
    — at memory bandwidth, no temporary; any edit of one element changes it, and so does a swap of two unequal elements (an
    in-place row shuffle).  The NumPy form below is the same function (used when the library cannot be loaded)."""
    w = np.ascontiguousarray(Z).reshape(-1).view(np.uint8)
    try:                     # one threaded pass in the library (host-only entry point: no GPU needed) — 32 MB of row weights in
        lib = _lib.load()    # ~1 ms where NumPy reductions take ~3 ms, as much as the pass they guard
    except _lib.MimoHipError:
        lib = None
    if lib is not None:
        out = (C.c_uint64 * 2)()
        if lib.mimo_host_checksum(w.ctypes.data_as(C.c_void_p), w.size, out) == 0:
            return (int(out[0]), int(out[1]))
    return _word_checksum_numpy(w)


def _word_checksum_numpy(w):
    n8 = w.size // 8
    u = w[:8 * n8].view(np.uint64)
    tail = w[8 * n8:]
    nw = n8 + (1 if tail.size else 0)
    m = u ^ (u >> np.uint64(32))
    mask = (1 << 64) - 1
    A = int(np.add.reduce(m, dtype=np.uint64)) if n8 else 0
    B = int(np.add.reduce(m * (np.uint64(nw) - np.arange(n8, dtype=np.uint64)), dtype=np.uint64)) if n8 else 0
    if tail.size:
        t = int.from_bytes(tail.tobytes(), 'little')
        mt = t ^ (t >> 32)
        A, B = (A + mt) & mask, (B + mt) & mask
    return (A, B)


class FrozenWeights:
    """Row weights a driver passes unchanged to every iteration of ONE call (hgmm.py:199-207: the outer responsibilities of an
    inner mixture): the content fingerprint — every byte — is taken once here instead of once per pass (32 MB of weights at
    N = 4e6 hash in ~1.3 ms, twice per pass: more than the pass itself).  Behaves like the array (np.asarray works)."""
    __slots__ = ('array', 'key')

    def __init__(self, w):
        self.array = np.ascontiguousarray(np.asarray(w, dtype=np.float64)).reshape(-1)
        self.key = (self.array.__array_interface__['data'][0], self.array.shape[0], content_fingerprint(self.array, exact=True))

    def __array__(self, dtype=None, copy=None):
        return self.array if dtype is None else self.array.astype(dtype, copy=False)

    def __len__(self):
        return self.array.shape[0]

    def __getitem__(self, idx):
        return self.array[idx]


def freeze_weights(w):
    """None stays None, a FrozenWeights is returned as it is, anything else is wrapped (fingerprint taken now)."""
    return w if w is None or isinstance(w, FrozenWeights) else FrozenWeights(w)


def _weights_key(w):
    """(array, identity of its content): FrozenWeights carry theirs, a plain array is hashed — every byte — now."""
    if isinstance(w, FrozenWeights):
        return w.array, w.key
    wv = _f64(w).reshape(-1)
    return wv, (wv.__array_interface__['data'][0], wv.shape[0], content_fingerprint(wv, exact=True))


def content_fingerprint(Z, exact=None):
    """Fingerprint of an array's CONTENT, so that an in-place edit between two calls (centring, whitening,
    a reused buffer filled with the next data set — the reference re-reads its arguments on every call, and itself
    edits caller arrays in place, gaussian.py:513) is seen and the device copy refreshed.  Arrays up to 2 MB: CRC of
    every byte; up to 16 MB: a checksum over every 8-byte word (sees any single edit, a few ms).  Beyond that the
    default is a sample — CRC of ~8192 evenly strided elements plus the first and last rows: O(1), catches every edit that
    touches the whole array or a contiguous block of it, not a change of a few isolated elements — unless `exact` (or the
    environment variable MIMO_BIND_EXACT=1) asks for the full word checksum (0.2 s at 1.28 GB: once per driver call, not
    per sweep).  engine.unbind() forces a re-upload whatever the fingerprint says."""
    import zlib
    Z = np.asarray(Z)
    if Z.nbytes <= _FULL_HASH_BYTES:
        return zlib.crc32(np.ascontiguousarray(Z).view(np.uint8).reshape(-1))
    if Z.flags.c_contiguous and (Z.nbytes <= _FULL_SUM_BYTES or (_exact_mode() if exact is None else exact)):
        return _word_checksum(Z)
    if Z.flags.c_contiguous:
        flat = Z.reshape(-1)
        step = max(1, flat.shape[0] // _SAMPLE_ELEMS)
        h = zlib.crc32(np.ascontiguousarray(flat[::step]).view(np.uint8))
    else:           # a strided view (a column block of a wider matrix): whole rows at even spacing, never a full copy
        n = Z.shape[0]
        per_row = max(1, int(np.prod(Z.shape[1:])))
        idx = np.unique(np.linspace(0, n - 1, num=min(n, max(2, _SAMPLE_ELEMS // per_row))).astype(np.int64))
        h = zlib.crc32(np.ascontiguousarray(Z[idx]).view(np.uint8).reshape(-1))
    h = zlib.crc32(np.ascontiguousarray(Z[0]).view(np.uint8).reshape(-1), h)
    return zlib.crc32(np.ascontiguousarray(Z[-1]).view(np.uint8).reshape(-1), h)


def _bind_key(Z, exact=None):
    return (Z.__array_interface__['data'][0], Z.shape, Z.strides, Z.dtype.str, content_fingerprint(Z, exact))


def bind(engine, Z, structure='full', exact=None):
    """Make `Z` ((N,Dz) float64 host array) the engine's resident data set, uploading it only if it is not the
    array bound last — identity = address + shape + a content fingerprint (`content_fingerprint`: exact up to 16 MB,
    sampled beyond unless `exact=True` / MIMO_BIND_EXACT=1), so an in-place edit of the bound array re-uploads it and
    drops the cached sum z z' — and select the structure of the precision blocks the caller is going to pass."""
    if hasattr(engine, 'set_structure'):
        engine.set_structure(structure)
    Z = np.asarray(Z)
    if Z.ndim == 1:
        Z = Z.reshape(-1, 1)
    key = _bind_key(Z, exact)
    if getattr(engine, "_bound_key", None) != key:
        engine.upload(Z)                 # (also invalidates the cached pooled second moment)
        engine._bound_key = key
        engine._bound_ref = Z        # keeps the address from being recycled
    elif _verify_mode() and isinstance(engine, BoundDataGuard) and Z.flags.c_contiguous and Z.dtype == np.float64 \
            and Z.nbytes > _FULL_SUM_BYTES and not (_exact_mode() if exact is None else exact) and engine._verify is None:
        # the sampled fingerprint matched: every byte is compared behind the first pass of this call (BoundDataGuard)
        engine._start_verify(Z)
    return engine


def unbind(engine):
    engine._bound_key = None
    engine._bound_ref = None


_default_engine = None


def default_engine():
    """Process-wide engine on device LOCAL_RANK (one process per GPU); raises without HIP."""
    global _default_engine
    if _default_engine is None:
        import os
        _default_engine = HipEngine(int(os.environ.get("LOCAL_RANK", "0")))
    return _default_engine


def set_default_engine(engine):
    global _default_engine
    _default_engine = engine
