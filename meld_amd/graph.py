"""Device-resident alpha-decay kNN graph: the MI355X replacement of ``graphtools.Graph``.

The reference builds its graph with ``graphtools.Graph(X, knn, decay, thresh, anisotropy=1,
use_pygsp=True, ...)`` (reference ``meld/meld.py:117-118,273``) and hands a PyGSP graph to
``meld/filter.py``.  ``DeviceGraph`` offers the attributes that path touches -- ``N``
(``meld/meld.py:209``), ``estimate_lmax()`` / ``lmax`` (``meld/filter.py:39,45``) -- plus host
exports (``W``, ``K``, ``L``, ``dw``) for inspection and tests.  All arithmetic is done by the HIP
kernels of ``libmeld_hip.so``; torch only owns the device memory and the stream.
"""
from __future__ import annotations

from ._options import is_set, opt

import math
import os
import sys
import time

import numpy as np
import torch

from ._lib import check, get_lib, ptr

__all__ = ["DeviceGraph", "build_knn_graph", "default_ksel", "EVENTS", "record_events"]

# Optional HIP-event timing of the hot kernels (bench.py turns it on).  Events are recorded on
# the stream the kernels are launched on (torch's current stream).
EVENTS = {"enabled": False, "pairs": []}


def record_events(enabled=True):
    EVENTS["enabled"] = bool(enabled)
    EVENTS["pairs"] = []


class _EventSpan:
    def __init__(self, name, **meta):
        self.name, self.meta = name, meta

    def __enter__(self):
        if EVENTS["enabled"]:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if EVENTS["enabled"]:
            self.b.record()
            EVENTS["pairs"].append((self.name, self.a, self.b, self.meta))
        return False


def event_times_ms():
    """name -> list of elapsed ms (call after a device sync)."""
    out = {}
    for name, a, b, meta in EVENTS["pairs"]:
        out.setdefault(name, []).append(a.elapsed_time(b))
    return out


def _stream():
    return torch.cuda.current_stream().cuda_stream


def default_ksel(knn):
    """Candidate-list length of the search kernel: ~4x the (knn+1) nearest, in [32, 128].
    (graphtools searches 6*(knn+1) first [UPSTREAM search_multiplier]; rows that need more go
    through the exact radius sweep exactly as graphtools re-searches them.)"""
    k = 4 * (knn + 1)
    k = ((k + 31) // 32) * 32
    return int(min(128, max(32, k)))


class _Timer:
    """Per-stage wall times (host clock around a device sync); only active when asked for."""

    def __init__(self, enabled):
        self.enabled = enabled
        self.t = {}
        self._t0 = None

    def start(self):
        if self.enabled:
            torch.cuda.synchronize()
            self._t0 = time.perf_counter()

    def stop(self, name):
        if self.enabled:
            torch.cuda.synchronize()
            self.t[name] = self.t.get(name, 0.0) + time.perf_counter() - self._t0
            self._t0 = time.perf_counter()


class _HookList(list):
    """Callables run once by the next directed_kernel_coo OF THIS THREAD right after it has launched the candidate search
    (fit_transform's label factorisation).  Per thread: two builds running concurrently must not run each other's hooks."""


class _PerThreadHooks:
    def __init__(self):
        import threading

        self._tls = threading.local()

    def _lst(self):
        lst = getattr(self._tls, "hooks", None)
        if lst is None:
            lst = self._tls.hooks = _HookList()
        return lst

    def append(self, h):
        self._lst().append(h)

    def remove(self, h):
        self._lst().remove(h)

    def pop(self):
        return self._lst().pop()

    def __contains__(self, h):
        return h in self._lst()

    def __bool__(self):
        return bool(self._lst())

    def __len__(self):
        return len(self._lst())


_WHILE_SEARCHING = _PerThreadHooks()


class DeviceGraph:
    """Symmetric weight matrix W (CSR, fp64 values, int32 columns, no diagonal) and degrees
    ``dw = W 1`` resident in HBM; ``L = diag(dw) - W`` is applied on the fly, never stored.

    Attributes mirror what callers of the PyGSP graph use: ``N``, ``lmax``, ``estimate_lmax()``,
    and (host-side, lazily copied) ``W``, ``K``, ``L``, ``dw``.
    """

    def __init__(self, rowptr, col, val, dw, ksum=None, anisotropy=1.0, row_begin=0, n_total=None, info=None):
        self.rowptr = rowptr  # int64 [n_rows + 1]
        self.col = col  # int32 [nnz]
        self.val = val  # fp64  [nnz]
        self.dw_dev = dw  # fp64  [n_rows]
        self.ksum = ksum  # fp64  [N] row sums of the symmetrised kernel (diag included)
        self.anisotropy = float(anisotropy)
        self.n_rows = int(rowptr.shape[0] - 1)
        self.row_begin = int(row_begin)
        self.N = int(n_total if n_total is not None else self.n_rows)
        self.nnz = int(col.shape[0])
        self.info = info or {}
        self._lmax = None
        self.lmax_info = {}
        # sharding: a single-GPU graph is one shard that owns every row
        self.rows_pad = self.n_rows  # local rows incl. padding (equal on every rank)
        self.n_pad = self.N  # global rows incl. padding
        self.comm = None
        self.ops = None
        # cache-locality permutation (device int64, new -> old index) or None: the device arrays are
        # in the permuted order, every host-facing export and the filter input/output use the
        # caller's original order
        self.perm = None
        self.bandwidth = None
        self.n_landmark = None

    @classmethod
    def from_scipy(cls, W, device="cuda"):
        """Upload a symmetric, zero-diagonal scipy.sparse weight matrix (e.g. ``G.W`` of a graph
        built elsewhere) -- the counterpart of handing a prebuilt graph to ``MELD.fit``
        (reference ``meld/benchmark.py:194-195``)."""
        from scipy import sparse

        W = sparse.csr_matrix(W).astype(np.float64)
        W.sort_indices()
        if W.shape[0] != W.shape[1]:
            raise ValueError("W must be square")
        rowptr = torch.from_numpy(W.indptr.astype(np.int64)).to(device)
        col = torch.from_numpy(W.indices.astype(np.int32)).to(device)
        val = torch.from_numpy(W.data.astype(np.float64)).to(device)
        dw = torch.from_numpy(np.ravel(W.sum(1)).astype(np.float64)).to(device)
        return cls(rowptr, col, val, dw, ksum=None, anisotropy=0.0)

    @classmethod
    def from_foreign(cls, G, W=None, device="cuda"):
        """Adopt a graph object built elsewhere (graphtools / pygsp: anything with a square
        scipy-sparse ``.W``): upload its weights, keep the spectral bound it already carries
        (pygsp caches ``estimate_lmax`` in ``_lmax``; its ``estimate_lmax`` is then a no-op, and so
        is ours) and, when it exposes graphtools' kernel ``.K``, the kernel's diagonal (used by
        ``VertexFrequencyCluster``'s diffusion operator)."""
        from scipy import sparse

        if not torch.cuda.is_available():
            raise RuntimeError("meld_amd needs a ROCm GPU (MI355X); there is no CPU fallback")
        W = sparse.csr_matrix(G.W if W is None else W)
        if W.diagonal().any():
            W = (W - sparse.diags(W.diagonal(), 0)).tocsr()  # weights carry no self loops (pygsp's W)
            W.eliminate_zeros()
        out = cls.from_scipy(W, device=device)
        lm = getattr(G, "_lmax", None)
        if lm is None and isinstance(getattr(type(G), "lmax", None), property) is False:
            lm = getattr(G, "lmax", None)  # a plain attribute (not pygsp's computing property)
        if lm is not None and np.isfinite(lm) and lm > 0:
            out.lmax = float(lm)
        try:
            K = getattr(G, "K", None)
        except Exception:
            K = None
        if K is not None and sparse.issparse(K) and K.shape == W.shape:
            out._kdiag = torch.from_numpy(np.ascontiguousarray(K.diagonal(), dtype=np.float64)).to(device)
        out.info["adopted_from"] = type(G).__name__
        return out

    # -- pygsp-like surface -------------------------------------------------------------------
    @property
    def lmax(self):
        if self._lmax is None:
            self.estimate_lmax()
        return self._lmax

    @lmax.setter
    def lmax(self, value):
        self._lmax = None if value is None else float(value)
        self.lmax_info = {} if value is None else dict(method="injected")

    def estimate_lmax(self, recompute=False, tol=1e-3, max_iter=300, method=None):
        """Largest Laplacian eigenvalue x 1.01 (pygsp's safety factor, [UPSTREAM pygsp
        ``Graph.estimate_lmax``] at reference ``meld/filter.py:39``).  pygsp stops ARPACK at
        tol=5e-3, which makes its value run-to-run noisy at the 1e-4 level; here a Lanczos
        recurrence on the device SpMV is run to a relative Ritz residual ``tol``: measured on the 1M-cell
        benchmark graph the eigenvalue error is 1.2e-6 relative at the default 1e-3 (35 iterations), 9e-8 at 3e-4
        (40), 4e-9 at 1e-4 (45) -- the error goes with the square of the residual; 1.2e-6 is 1/40 of the spread of
        the reference's own estimate between two runs.  On the panel-tiled layout the SpMV streams an fp32 copy of
        the weights (vectors and sums fp64): that moves the eigenvalue by < 1e-7 relative.  No-op when a value is
        already set (same as pygsp), which is how parity tests inject a common lmax."""
        if self._lmax is not None and not recompute:
            # a value that is already there is kept (pygsp's no-op) -- unless it came from the OTHER estimator: asking for
            # the reference's ARPACK number after a Lanczos estimate (or the reverse) computes it
            have = self.lmax_info.get("method", "injected")
            if method is None or have == "injected" or have == method:
                return self._lmax
        method = method or "lanczos"
        if method == "arpack":
            return self._estimate_lmax_arpack()
        if method != "lanczos":
            raise ValueError("lmax method {!r} not recognized. Choose from ['lanczos', 'arpack']".format(method))
        from .filter import lanczos_lmax

        lam, info = lanczos_lmax(self, tol=tol, max_iter=max_iter)
        self._lmax = 1.01 * lam
        self.lmax_info = dict(info, method="lanczos")
        return self._lmax

    def _estimate_lmax_arpack(self):
        """The reference's own estimate, call for call ([UPSTREAM pygsp 0.5.1 ``Graph.estimate_lmax``] at
        reference ``meld/filter.py:39``): the Laplacian is copied to the host once and
        ``1.01 * scipy.sparse.linalg.eigsh(L, k=1, tol=5e-3, ncv=min(N, 10))`` is evaluated there
        (``2 max(dw)`` if ARPACK does not converge).  Opt-in (``MELD(lmax="arpack")``): slower than the
        device Lanczos and only converged to ARPACK's 5e-3, but it is the number the reference stack
        computes -- with it an un-injected ``fit_transform`` meets the reference within its own
        run-to-run spread (ARPACK's start vector comes from process-global state)."""
        from scipy.sparse.linalg import ArpackNoConvergence, eigsh

        if self.n_rows != self.N:
            raise NotImplementedError("lmax='arpack' needs an unsharded graph")
        L = self.L
        try:
            lam = eigsh(L, k=1, tol=5e-3, ncv=min(self.N, 10), return_eigenvectors=False)
            self._lmax = 1.01 * float(lam[0])
            self.lmax_info = dict(method="arpack", tol=5e-3)
        except ArpackNoConvergence:  # pragma: no cover
            self._lmax = 2.0 * float(np.max(self.dw))
            self.lmax_info = dict(method="arpack", converged=False)
        return self._lmax

    # -- host exports in the caller's cell order (tests / inspection; not used by the hot path) -----
    def _inv_perm_host(self):
        if self.perm is None:
            return None
        p = self.perm.cpu().numpy()
        inv = np.empty_like(p)
        inv[p] = np.arange(p.shape[0])
        return inv

    def _scipy(self, vals):
        from scipy import sparse

        M = sparse.csr_matrix((vals, self.col.cpu().numpy(), self.rowptr.cpu().numpy()[: self.n_rows + 1]), shape=(self.n_rows, self.N))
        inv = self._inv_perm_host()
        if inv is not None:
            if self.n_rows != self.N:
                raise ValueError("host export of a sharded, permuted graph is not supported")
            M = M[inv][:, inv].tocsr()
            M.sort_indices()
        return M

    def _vec_host(self, t):
        v = t.cpu().numpy()
        inv = self._inv_perm_host()
        return v if inv is None or v.shape[0] != inv.shape[0] else v[inv]

    @property
    def W(self):
        return self._scipy(self.val.cpu().numpy())

    @property
    def dw(self):
        return self._vec_host(self.dw_dev)

    @property
    def bandwidth_host(self):
        """bandwidths in the units of the graph's metric (``bandwidth`` itself is the euclidean one of the rows the search saw)"""
        if getattr(self, "bandwidth", None) is None:  # (a precomputed affinity has none)
            return None
        to_metric = getattr(self, "bandwidth_to_metric", None)
        return self._vec_host(self.bandwidth if to_metric is None else to_metric(self.bandwidth))

    @property
    def L(self):
        from scipy import sparse

        if self.n_rows != self.N:
            raise ValueError("L is only defined for an unsharded graph")
        return (sparse.diags(self.dw, 0) - self.W).tocsr()

    @property
    def K(self):
        """Symmetrised, anisotropy-normalised kernel including its diagonal (graphtools' ``G.K``)."""
        from scipy import sparse

        if self.n_rows != self.N:
            raise ValueError("K is only defined for an unsharded graph")
        if getattr(self, "_kdiag", None) is not None:  # dense graph: the diagonal was computed explicitly
            return (self.W + sparse.diags(self._vec_host(self._kdiag), 0)).tocsr()
        return (self.W + sparse.diags(self._vec_host(self.kernel_diagonal()), 0)).tocsr()

    # -- what the reference's side paths read from the graph: graphtools' ``knn`` / ``diff_op``
    # (reference meld/cluster.py:213, comparison/comparison.py:318-323), pygsp's Fourier basis (meld/cluster.py:235-236) --
    @property
    def knn(self):
        """The ``knn`` the graph was built with (graphtools' attribute; None for a weight matrix uploaded from elsewhere)."""
        return self.info.get("knn")

    @property
    def diff_op(self):
        """graphtools' diffusion operator: the kernel (diagonal included) with rows normalised to sum 1, as scipy CSR in
        the caller's cell order (built on first use, on the host: an inspection / side-path export like ``K``)."""
        if getattr(self, "_diff_op", None) is None:
            from scipy import sparse

            K = self.K
            self._diff_op = (sparse.diags(1.0 / np.ravel(K.sum(1)), 0) @ K).tocsr()
        return self._diff_op

    FOURIER_MAX_N = 16384

    def compute_fourier_basis(self, recompute=False):
        """pygsp's ``compute_fourier_basis``: full eigendecomposition of the combinatorial Laplacian -- ``e`` ascending,
        ``U`` the eigenvectors (columns), both host arrays in the caller's cell order; also sets ``lmax = e[-1]`` like
        pygsp.  Dense O(N^3) on the device (rocSOLVER ``eigh``), offered up to FOURIER_MAX_N cells."""
        if getattr(self, "_U", None) is not None and not recompute:
            return
        if self.n_rows != self.N:
            raise ValueError("the Fourier basis is only defined for an unsharded graph")
        if self.N > self.FOURIER_MAX_N:
            raise NotImplementedError("compute_fourier_basis is a dense O(N^3) eigendecomposition; N={} exceeds the {} cells "
                                      "it is offered for".format(self.N, self.FOURIER_MAX_N))
        n, dev = self.N, self.val.device
        row_of = torch.repeat_interleave(torch.arange(n, device=dev), self.rowptr[1:] - self.rowptr[:-1])
        L = torch.zeros(n, n, dtype=torch.float64, device=dev)
        L[row_of, self.col.to(torch.int64)] = -self.val
        L += torch.diag(self.dw_dev[:n])
        e, U = torch.linalg.eigh(L)
        e, U = e.cpu().numpy(), U.cpu().numpy()
        inv = self._inv_perm_host()
        if inv is not None:
            U = U[inv]
        self._e, self._U = e, U
        self.lmax = float(e[-1])

    @property
    def U(self):
        self.compute_fourier_basis()
        return self._U

    @property
    def e(self):
        self.compute_fourier_basis()
        return self._e

    @property
    def landmark_op(self):
        """graphtools' LandmarkGraph operator (reference ``meld/meld.py:105`` forwards ``n_landmark``): lazily built
        there, never used by MELD's filter; not implemented here."""
        raise NotImplementedError("the landmark operator (graphtools LandmarkGraph.landmark_op) is not implemented; "
                                  "MELD's density estimate does not use it")

    def kernel_diagonal(self):
        """Diagonal of the kernel matrix K in the device order: K_ii = 1 / (ksum_i^2)^anisotropy for a
        graph built here (the alpha-decay kernel has 1 on its diagonal before the anisotropy
        normalisation); 1 for an uploaded weight matrix that came without its kernel
        (``from_scipy``: the kernel's own diagonal before any normalisation)."""
        if getattr(self, "_kdiag", None) is not None:
            return self._kdiag
        if self.ksum is None:
            return torch.ones(self.N, dtype=torch.float64, device=self.val.device)
        return 1.0 / (self.ksum * self.ksum) ** self.anisotropy


_BLAS_CTL = None


def _eigh_one_thread(a):
    """``np.linalg.eigh(a, UPLO="U")`` (row-major upper triangle) on one BLAS thread (threadpoolctl; plain numpy where it is absent)."""
    global _BLAS_CTL
    if a.shape[0] <= 64:  # (below OpenBLAS's threading threshold: 0.1 ms at 50 x 50 as it is)
        return np.linalg.eigh(a, UPLO="U")
    try:
        if _BLAS_CTL is None:
            from threadpoolctl import ThreadpoolController

            _BLAS_CTL = ThreadpoolController()
        with _BLAS_CTL.limit(limits=1, user_api="blas"):
            return np.linalg.eigh(a, UPLO="U")
    except ImportError:
        return np.linalg.eigh(a, UPLO="U")


def _scan_i32(lib, x, st):
    n = x.shape[0]
    out = torch.empty(n + 1, dtype=torch.int64, device=x.device)
    tb = lib.meld_scan_temp_bytes(n)
    tmp = torch.empty(tb, dtype=torch.uint8, device=x.device)
    check(lib.meld_exclusive_scan_i32_i64(ptr(x), ptr(out), n, ptr(tmp), tb, st), "meld_exclusive_scan_i32_i64")
    return out


class HipOps:
    """Local (per-GPU) stages of the path as calls into libmeld_hip.so.  The single-GPU builder
    and the row-sharded driver (``meld_amd.distributed``) compose the same stages; only the
    exchanges between them differ."""

    name = "hip"

    def __init__(self, device=None, search=None, prune=None, nprod=None, spmm=None):
        self.lib = get_lib()
        # recurrence kernel: "auto" (panel-tiled layout for graphs of at least PT_MIN_ROWS local rows),
        # "tiled" (always), "csr" (the CSR-stream kernel of spmm.hip)
        self.spmm = spmm
        # precision of the f16x3 search on the coordinate K blocks: 1 = fp16 hi parts only (half the MFMAs,
        # error bound 2^-9 max|x|^2), 3 = full hi/lo split (2^-16).  Either way the result is exact: rows
        # the bound cannot certify go through the exact sweep.
        self.nprod = int(opt("MELD_KNN_NPROD", "1")) if nprod is None else int(nprod)
        # exact tile pruning in the f16x3 search: off by default -- on the 10-d-intrinsic benchmark mixture
        # the bounding spheres of 64-cell tiles (radius 0.72) dwarf the neighbour radius (0.63), 98 % of
        # the (workgroup, tile) pairs stay live and the table costs 3 ms; it pays on low-dimensional or
        # well-separated data
        self.prune = (opt("MELD_KNN_PRUNE", "1") != "0") if prune is None else bool(prune)
        self.radius_cut = opt("MELD_KNN_RADIUS_CUT", "1") != "0"
        # thresholds of the first pass seeded from every row's own block (meld_knn16_seed_thresholds)
        self.seed = opt("MELD_KNN_SEED", "1") != "0"
        # per-query test of the pruning table against those seeds (meld_knn16_bounds, thr_seed)
        self.seeded_bounds = opt("MELD_KNN_SEEDED_BOUNDS", "1") != "0"
        # pruned search: query blocks dispatched by decreasing work (meld_knn16_block_work)
        self.block_order = opt("MELD_KNN_BLOCK_ORDER", "1") != "0"
        # the first pass walks precomputed step lists (meld_knn16_step_lists) instead of testing the pruning table step by step
        self.step_lists = opt("MELD_KNN_STEP_LISTS", "1") != "0"
        # the search runs in the cells' principal frame where that concentrates the distances in the leading coordinates (the
        # list-driven first pass tests a block behind its first K block: principal_frame)
        self.rotate = opt("MELD_KNN_ROTATE", "1") != "0"
        # ... from this many cells on: the frame costs ~0.9 ms whatever the size (a read-back and a 50 x 50 eigenproblem on the host
        # among it) and pays from ~250k cells (200k: 10.0 vs 9.8 ms per step without it; 350k: 14.6 vs 15.4; 500k: 21.0 vs 22.5)
        self.rotate_min_cells = int(opt("MELD_KNN_ROTATE_MIN", "262144"))
        # candidate-search kernel: "f16x3" (split-fp16 MFMA) or "f32" (fp32 MFMA)
        self.search = search or opt("MELD_KNN_SEARCH", "f16x3")
        if self.search not in ("f16x3", "f32"):
            raise ValueError("unknown search kernel {!r}".format(self.search))
        if not torch.cuda.is_available():
            raise RuntimeError("meld_amd needs a ROCm GPU (MI355X); there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)

    def col_stats(self, X):
        """(sums, minima, maxima) of the columns of an fp64 [N, d <= 256] matrix in one pass (``meld_col_stats_f64``)."""
        N, d = int(X.shape[0]), int(X.shape[1])
        out = torch.empty(3, d, dtype=torch.float64, device=X.device)
        tb = self.lib.meld_col_stats_temp_bytes(d)
        tmp = torch.empty(tb, dtype=torch.uint8, device=X.device)
        check(self.lib.meld_col_stats_f64(ptr(X), N, d, ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(tmp), tb, _stream()), "meld_col_stats_f64")
        return out[0], out[1], out[2]

    def frame_axes_async(self, X, col_stats):
        """Start the frame of the search on ``X`` BEFORE the cells are brought into locality order: the covariance of evenly spaced
        rows does not depend on the order of the cells, so its scatter matrix and the read-back go out first, and the d x d
        eigenproblem is solved on the host while the GPU works through the ordering (the GPU used to idle ~0.35 ms at 1M cells
        while the host waited for the read-back, decomposed the matrix and sent the axes back).  Returns a function that finishes
        the job -- ``finish(lead)`` -> the axes as rows ``At`` (device, [d, meld_frame_max_dims()]) or None (frame declined) -- or
        None where the frame will not be asked for (options, size, width) or the column sums are not at hand."""
        lib = self.lib
        N, d = int(X.shape[0]), int(X.shape[1])
        if (col_stats is None or opt("MELD_FRAME_ASYNC", "1") == "0" or not self.rotate or not self.prune or not self.step_lists or not self.seed or self.nprod != 1 or self.search != "f16x3"
                or N < max(16384, self.rotate_min_cells) or d > int(lib.meld_frame_max_dims()) or int(lib.meld_knn16_split_dims(d)) <= 0):
            return None
        st = _stream()
        mean = col_stats[0] / N
        cov = torch.zeros(d, d, dtype=torch.float64, device=X.device)
        check(lib.meld_cov_sample_f64(ptr(X), N, d, ptr(mean), max(1, N // 32768), ptr(cov), st), "meld_cov_sample_f64")
        host = torch.empty(d, d, dtype=torch.float64, pin_memory=True)
        host.copy_(cov, non_blocking=True)
        done = torch.cuda.Event()
        done.record()

        def finish(lead):
            done.synchronize()
            evals, evecs = _eigh_one_thread(host.numpy())
            tot = float(evals.sum())
            if not np.isfinite(tot) or tot <= 0.0 or float(evals[-lead:].sum()) < 0.5 * tot:
                return False
            At = np.zeros((d, int(lib.meld_frame_max_dims())))
            At[:, :d] = evecs[:, ::-1].T
            return torch.from_numpy(At).to(X.device, non_blocking=True)

        return finish

    def principal_frame(self, X, mean, lead, comm=None, axes=None):
        """The cells in their principal frame, ``(X - mean) V`` with the eigenvectors of the covariance (of at most 32768 evenly
        spaced rows, ``meld_cov_sample_f64``) in descending order of variance (``meld_rotate_rows_f64``) -- or None when the
        ``lead`` leading coordinates would carry less than half of the variance (the search's partial test would seldom drop a
        block; any frame is valid, the graph is the same).  Distances are those of ``X`` to
        rounding (V is orthonormal to 1e-15): the search only nominates candidates, their distances are evaluated in fp64 on
        ``X`` itself.  The d x d eigenproblem is solved on the host (one read-back of d x d numbers)."""
        lib, st = self.lib, _stream()
        N, d = int(X.shape[0]), int(X.shape[1])
        if axes is not None and (comm is None or getattr(comm, "world", 1) == 1):
            # (the axes were started before the ordering: frame_axes_async)
            At = axes(lead)
            if At is False:
                return None
            out = torch.empty_like(X)
            check(lib.meld_rotate_rows_f64(ptr(X), N, d, ptr(mean), ptr(At), ptr(out), st), "meld_rotate_rows_f64")
            return out
        wide = d > int(lib.meld_frame_max_dims())  # (beyond the frame kernels' row width: the same two products through the library)
        stride = max(1, N // 32768)
        if wide:
            Xs = X[::stride] - mean
            cov = torch.triu(Xs.T @ Xs)
            del Xs
        else:
            cov = torch.zeros(d, d, dtype=torch.float64, device=X.device)
            check(lib.meld_cov_sample_f64(ptr(X), N, d, ptr(mean), stride, ptr(cov), st), "meld_cov_sample_f64")
        if comm is not None and getattr(comm, "world", 1) > 1:
            # ranks of a row-sharded build hold the same cells but sum their scatter matrices in different orders (atomics): all
            # take rank 0's bits, so that the frame -- and the decision to use one -- is the same everywhere (the tile spheres
            # are shared between the ranks)
            if comm.rank != 0:
                cov.zero_()
            comm.all_reduce_sum(cov)
        # (ONE BLAS thread for the d x d eigenproblem: past ~64 x 64 OpenBLAS goes multi-threaded, and on a many-core host its
        # threads' start-up cost 60-370 ms for a 100 x 100 matrix that one thread decomposes in 1.1 ms -- found as a 62 ms idle gap
        # of the GPU in the d = 100 step)
        evals, evecs = _eigh_one_thread(cov.cpu().numpy())
        tot = float(evals.sum())
        if not np.isfinite(tot) or tot <= 0.0 or float(evals[-lead:].sum()) < 0.5 * tot:
            return None
        if wide:
            # d in (64, 141]: a plain N x d x d fp64 GEMM (rocBLAS) -- PCA scores, the reference's default input (n_pca = 100), are
            # already in their principal frame up to this rotation's rounding, so it is close to a permutation there
            V = torch.from_numpy(np.ascontiguousarray(evecs[:, ::-1])).to(X.device)
            return torch.addmm(-(mean @ V), X, V)
        At = np.zeros((d, int(lib.meld_frame_max_dims())))
        At[:, :d] = evecs[:, ::-1].T  # the axes as rows, by descending variance
        At = torch.from_numpy(At).to(X.device)
        out = torch.empty_like(X)
        check(lib.meld_rotate_rows_f64(ptr(X), N, d, ptr(mean), ptr(At), ptr(out), st), "meld_rotate_rows_f64")
        return out

    # ---- A2 + A3: directed alpha-decay kernel rows of [q_begin, q_begin + q_count) as COO -------
    def directed_kernel_coo(self, X, q_begin, q_count, knn, decay, thresh, ksel, tm=None, force_fallback=False, n_refs=None, assemble=False, comm=None,
                            bw_scale=1.0, bw_fixed=None, col_stats=None, knn_max=None, symm=(0, 0.0), count_rows_ge=None, frame_axes=None):
        """Returns (keys[2M] int64, vals[2M] fp64, info): slot e < M holds (i, j, K_ij / 2) with
        key = i << 32 | j for the local row i; slot M + e holds the transposed (j, i, K_ij / 2).

        ``n_refs``: search BETWEEN two point sets (the cross blocks of the MNN kernel, ``meld_amd.mnn``): the references
        are the rows [0, n_refs) of X only, the queries lie behind them, there is no self among a row's candidates --
        the caller passes knn - 1 so that the bandwidth is the knn-th nearest reference.  No pruning table, no seeds
        (both rest on tile = query block).

        ``bw_scale`` / ``bw_fixed`` ([UPSTREAM graphtools kNNGraph ``bandwidth_scale`` / ``bandwidth``, forwarded by reference
        ``meld/meld.py:106,117-118``): the kernel uses ``max(bw * bw_scale, eps)``; ``bw_fixed`` (fp64 device tensor [N], in
        the order of ``X``) replaces the adaptive k-th-neighbour bandwidth -- the kernel radius of every row is then known
        before the search, which starts its thresholds there and cuts nothing on its own.  ``knn_max`` ([UPSTREAM kNNGraph
        ``knn_max``]): a row keeps its knn_max nearest cells (besides itself) at most.  ``count_rows_ge``: report the number of rows
        whose kernel radius holds at least that many cells, self counted (``info["rows_with_at_least"]``: what graphtools'
        re-search loop branches on, see ``build_knn_graph``)."""
        lib, st, dev = self.lib, _stream(), X.device
        tm = tm or _Timer(False)
        N, d = int(X.shape[0]), int(X.shape[1])
        cross = n_refs is not None
        NR = int(n_refs) if cross else N  # references of the search
        if cross and not (0 < NR <= q_begin and q_begin + q_count <= N):
            raise ValueError("cross search: the queries must lie behind the n_refs references")
        tm.start()
        col_min = col_max = None
        if col_stats is not None:  # (the front end's pass over X -- its NaN / infinity check -- already has them)
            sums, col_min, col_max = col_stats
            mean = sums / N
        elif d <= 256:
            # one pass: the mean and the columns' extremes (from which the operand scale follows without another pass over X)
            sums, col_min, col_max = self.col_stats(X)
            mean = sums / N
        else:  # beyond the column-sum kernel's width (only the library search path handles such data)
            mean = X.mean(dim=0)
        norm2 = torch.empty(N, dtype=torch.float32, device=dev)
        nmax = torch.zeros(1, dtype=torch.float32, device=dev)
        X_search, mean_search = X, mean  # (the f16x3 search may move to the cells' principal frame)
        search = self.search
        cand_thr, rfac, tiles_done = None, 1.0, None
        used_prune = used_seed = used_seeded_bounds = used_block_order = used_step_lists = used_two_phase = False
        partial_in_search = 0
        if search == "f16x3" and lib.meld_knn16_kblocks(d) < 0:
            search = "wide"  # d beyond the instantiated MFMA kernels (d > 141)
        if cross and search != "f16x3":
            raise NotImplementedError("the search between two point sets runs on the split-fp16 MFMA kernel only (d <= 141)")
        if search == "wide":
            # Library path for wide data that was not reduced by PCA: chunked fp64 GEMMs (rocBLAS) for
            # |q|^2 + |r|^2 - 2 q.r and torch.topk merges, feeding the same exact refinement.  No hand-written
            # kernel: the reference's default (n_pca = 100) never gets here, and the distance GEMM at d >> 100 is
            # a plain library GEMM.
            research = None
            cap = int(ksel)
            err_coef, err_lin = 1e-6, 0.0  # fp64 GEMM form + fp32 storage of d2: << 1e-6 max|x~|^2
            Xc = X - mean
            n2 = (Xc * Xc).sum(dim=1)
            norm2.copy_(n2.to(torch.float32))
            nmax.copy_(n2.max().to(torch.float32).reshape(1))
            tm.stop("prepare")
            kk = min(int(ksel), N)
            cand_idx = torch.zeros(q_count * cap, dtype=torch.int32, device=dev)
            cand_d2 = torch.full((q_count * cap,), float("inf"), dtype=torch.float32, device=dev)
            cand_cnt = torch.full((q_count,), kk, dtype=torch.int32, device=dev)
            QC, RC = 4096, 32768
            with _EventSpan("knn_topk", N=N, d=d, q=q_count):
                for q0 in range(0, q_count, QC):
                    q1 = min(q_count, q0 + QC)
                    Xq = Xc[q_begin + q0 : q_begin + q1]
                    nq = n2[q_begin + q0 : q_begin + q1]
                    best_d = torch.full((q1 - q0, 0), 0.0, dtype=torch.float64, device=dev)
                    best_i = torch.zeros((q1 - q0, 0), dtype=torch.int64, device=dev)
                    for r0 in range(0, N, RC):
                        r1 = min(N, r0 + RC)
                        D = nq[:, None] + n2[None, r0:r1] - 2.0 * (Xq @ Xc[r0:r1].T)
                        ids = torch.arange(r0, r1, device=dev, dtype=torch.int64)[None, :].expand(q1 - q0, -1)
                        D = torch.cat([best_d, D], dim=1)
                        ids = torch.cat([best_i, ids], dim=1)
                        best_d, sel = torch.topk(D, min(kk, D.shape[1]), dim=1, largest=False, sorted=True)
                        best_i = torch.gather(ids, 1, sel)
                    rows = torch.arange(q0, q1, device=dev, dtype=torch.int64)[:, None] * cap + torch.arange(kk, device=dev)[None, :]
                    cand_d2[rows.reshape(-1)] = best_d.clamp_(min=0.0).to(torch.float32).reshape(-1)
                    cand_idx[rows.reshape(-1)] = best_i.to(torch.int32).reshape(-1)
            KP = d
            Q = Rt = None
            del Xc
        elif search == "f16x3":
            # split-fp16 operands on v_mfma_f32_32x32x16_f16 (knn16.hip)
            # In very low dimension the neighbours are so close (relative to max|x|^2) that the fp16-hi first
            # pass certifies almost nothing (1M x 3: 968k of 1M rows re-searched) and is wasted; there the full
            # split costs next to nothing (one K block), so it is used from the start.  Same result either way.
            nprod = 3 if (self.nprod == 1 and d <= 6) else self.nprod
            KB = lib.meld_knn16_kblocks(d)
            if KB < 0:
                check(KB, "meld_knn16_kblocks")
            TS, BQ = lib.meld_knn16_tile_refs(), lib.meld_knn16_block_queries()
            cap = lib.meld_knn16_row_capacity(ksel)
            if cap < 0:
                check(cap, "meld_knn16_row_capacity")
            err_coef = lib.meld_knn16_error_coef_const(nprod, d)
            err_lin = lib.meld_knn16_error_coef_lin(nprod)
            n_tiles = (NR + TS - 1) // TS
            q_pad = ((q_count + BQ - 1) // BQ) * BQ
            # The frame of the search (X_s, its mean and column extremes): the cells' principal frame where the first pass can
            # use it -- operands in the split layout, seeds and step lists, the whole graph or a row shard of it -- else X itself.
            # Everything up to the candidate lists works on X_s; refinement and the exact sweeps on X.
            lead = int(lib.meld_knn16_split_dims(d))
            if (self.rotate and lead > 0 and nprod == 1 and not cross and self.prune and self.step_lists and self.seed
                    and N >= max(16384, self.rotate_min_cells) and q_begin % BQ == 0 and bw_fixed is None):
                X_s = self.principal_frame(X, mean, lead, comm, axes=frame_axes)
                if X_s is not None:
                    sums_s, col_min, col_max = self.col_stats(X_s)
                    X_search, mean_search = X_s, sums_s / N
            Rt = torch.empty(n_tiles * lib.meld_knn16_tile_bytes(d), dtype=torch.uint8, device=dev)
            Q = torch.empty(q_pad * lib.meld_knn16_query_bytes(d), dtype=torch.uint8, device=dev)
            Qn = torch.empty(q_pad, dtype=torch.float32, device=dev)
            scale_info = torch.empty(4, dtype=torch.float32, device=dev)
            if cross:
                check(lib.meld_knn16_prepare_cross(ptr(X), NR, N, d, ptr(mean), q_begin, q_count, ptr(Rt), ptr(Q), ptr(Qn), ptr(norm2), ptr(nmax), ptr(scale_info), st), "meld_knn16_prepare_cross")
                # the error bounds speak of the largest norm among ALL points of the search, and refine reads the
                # query's own norm at its row
                norm2[q_begin : q_begin + q_count] = Qn[:q_count]
                nmax = torch.maximum(nmax, Qn[:q_count].max().reshape(1))
            else:
                if col_min is not None:
                    check(lib.meld_knn16_prepare_scaled(ptr(X_search), N, d, ptr(mean_search), ptr(col_min), ptr(col_max), q_begin, q_count, ptr(Rt), ptr(Q), ptr(Qn), ptr(norm2), ptr(nmax), ptr(scale_info), st), "meld_knn16_prepare_scaled")
                else:
                    check(lib.meld_knn16_prepare(ptr(X_search), N, d, ptr(mean_search), q_begin, q_count, ptr(Rt), ptr(Q), ptr(Qn), ptr(norm2), ptr(nmax), ptr(scale_info), st), "meld_knn16_prepare")
            tm.stop("prepare")
            cand_idx = torch.empty(q_pad * cap, dtype=torch.int32, device=dev)
            cand_d2 = torch.empty(q_pad * cap, dtype=torch.float32, device=dev)
            cand_cnt = torch.empty(q_pad, dtype=torch.int32, device=dev)
            if self.radius_cut and knn < ksel:
                # rows are cut at the kernel radius their (knn+1)-th neighbour so far implies; the search
                # publishes each row's final threshold for refine's completeness test
                cand_thr = torch.full((q_pad,), float("inf"), dtype=torch.float32, device=dev)
                rfac = 1.0 if math.isinf(decay) else float((-math.log(thresh)) ** (1.0 / decay))
                # (the cut keeps everything within max(rf * bandwidth_scale, 1) bandwidths: never less than the bandwidth entry)
                rfac = max(rfac * float(bw_scale), 1.0)
            lb2 = block_order = step_list = step_cnt = None
            tiles_done = torch.zeros(4, dtype=torch.int64, device=dev)  # [(wave, tile) pairs looked at, blocks of 32 references past the partial test (two-pass route: pairs the search computed, then its blocks)]
            will_prune = self.prune and q_begin % TS == 0 and N >= 16384 and not cross
            n_blocks = q_pad // BQ
            seeds = None
            knn_cut = knn  # the radius cut of the search follows the knn-th neighbour ...
            if bw_fixed is not None and cand_thr is not None:
                # ... unless the bandwidth is given: every row's radius is known, the thresholds start there (scaled units, with
                # the row's search-error allowance on top) and the search cuts nothing itself (knn_cut = 0)
                knn_cut = 0
                rf_real = 1.0 if math.isinf(decay) else float((-math.log(thresh)) ** (1.0 / decay))
                rad = (bw_fixed[q_begin : q_begin + q_count] * float(bw_scale)).clamp_(min=float(np.finfo(float).eps)) * rf_real
                nmx = nmax.to(torch.float64)
                e_row = float(err_coef) * nmx + float(err_lin) * torch.sqrt(norm2[q_begin : q_begin + q_count].to(torch.float64) * nmx)
                seeds = torch.full((q_pad,), float("inf"), dtype=torch.float32, device=dev)
                seeds[:q_count] = ((rad * rad + 1.01 * e_row) * scale_info[0].to(torch.float64) ** 2 * (1.0 + 1e-5)).to(torch.float32)
                if q_pad > q_count:
                    seeds[q_count:] = seeds[q_count - 1]
            elif self.seed and cand_thr is not None and q_begin % BQ == 0 and not cross:
                # every row starts at the kernel radius its own block of BQ cells implies instead of at +inf
                seeds = torch.empty(q_pad, dtype=torch.float32, device=dev)
                if opt("MELD_KNN_SEED", "1") == "2":  # the fp32 kernel over the own block only
                    check(lib.meld_knn16_seed_thresholds(ptr(X_search), N, d, ptr(mean_search), ptr(scale_info), ptr(nmax), q_begin, q_count, knn, rfac, nprod, ptr(seeds), st), "meld_knn16_seed_thresholds")
                else:
                    check(lib.meld_knn16_seed_thresholds_mfma(ptr(Q), ptr(Qn), ptr(Rt), ptr(scale_info), ptr(nmax), N, d, q_begin, q_count, knn, rfac, nprod, int(opt("MELD_KNN_SEED_SIDE", "0")), ptr(seeds), st), "meld_knn16_seed_thresholds_mfma")
                tm.stop("seed")
                if opt("MELD_KNN_SEEDS_FROM"):  # (development: what perfect start thresholds would be worth -- the final thresholds of an earlier run)
                    seeds = torch.minimum(seeds, torch.load(opt("MELD_KNN_SEEDS_FROM")).to(dev))
            if will_prune:
                # (after the seeds: with them the table also drops the tiles no query of a wave can reach from
                # its own start threshold, see meld_knn16_bounds)
                tb = lib.meld_knn16_bounds_temp_bytes(N, d, q_count)
                tmpb = torch.empty(tb, dtype=torch.uint8, device=dev)
                spheres_shared = False
                if comm is not None and getattr(comm, "world", 1) > 1:
                    # row-sharded build: the spheres of the reference tiles are the same on every rank (0.7 ms at 1M cells):
                    # every rank computes 1 / world of them and the three arrays are all-gathered (4 MB in all)
                    import ctypes as C

                    rows_c, row_b = C.c_int64(0), C.c_int64(0)
                    check(lib.meld_knn16_sphere_layout(N, d, C.byref(rows_c), C.byref(row_b)), "meld_knn16_sphere_layout")
                    rows_c, row_b = int(rows_c.value), int(row_b.value)
                    if rows_c % comm.world == 0:
                        per = rows_c // comm.world
                        t0s = min(comm.rank * per, n_tiles)
                        t1s = min(t0s + per, n_tiles)
                        tmpb.zero_()
                        check(lib.meld_knn16_tile_spheres(ptr(X_search), N, d, ptr(mean_search), ptr(scale_info), ptr(tmpb), t0s, max(t1s - t0s, 0), st), "meld_knn16_tile_spheres")
                        parts = (tmpb[: rows_c * row_b], tmpb[rows_c * row_b : rows_c * (row_b + 4)], tmpb[rows_c * (row_b + 4) : rows_c * (row_b + 8)])
                        for arr, width in zip(parts, (row_b, 4, 4)):
                            mine = arr[comm.rank * per * width : (comm.rank + 1) * per * width].clone()
                            comm.all_gather_rows(arr, mine)
                        spheres_shared = True
                seeded_bounds = seeds is not None and self.seeded_bounds
                # (few query blocks -- a row shard, a mid-sized data set -- are searched in reference slices: a slice of a list-driven
                # launch walks every S-th entry of the block's list, MELD_KNN_LIST_SLICES=0 sends them to the table-driven kernel)
                resident_all = lib.meld_knn16_resident_blocks(d, nprod)
                few_blocks = resident_all > 0 and n_blocks < 2 * resident_all
                want_lists = self.step_lists and seeds is not None and cand_thr is not None and nprod == 1 \
                    and not (few_blocks and opt("MELD_KNN_LIST_SLICES", "1") == "0")
                direct = want_lists and seeded_bounds and not spheres_shared and q_begin == 0 and q_count == N and not cross \
                    and opt("MELD_KNN_LIST_DIRECT", "1") != "0" and not opt("MELD_KNN_SYMMETRIC_BOUNDS_OFF")
                if direct:
                    # queries = all the cells: the lists come straight from the cells (bounds as two bits per (wave, tile); the fp16
                    # table, its symmetrisation pass and the list builder's pass over it never exist)
                    step_list = torch.empty(n_blocks * n_tiles, dtype=torch.int32, device=dev)
                    step_cnt = torch.empty(n_blocks, dtype=torch.int32, device=dev)
                    scratch = torch.empty(lib.meld_knn16_list_scratch_bytes(N), dtype=torch.uint8, device=dev)
                    check(lib.meld_knn16_step_lists_direct_lead(ptr(X_search), N, d, ptr(mean_search), ptr(scale_info), ptr(nmax), ptr(Rt), ptr(seeds), ptr(Qn), nprod,
                                                                ptr(tmpb), ptr(scratch), ptr(step_list), n_tiles, ptr(step_cnt), int(X_search is not X), st), "meld_knn16_step_lists_direct")
                    del scratch
                    work = step_cnt
                else:
                    lb2 = torch.empty(lib.meld_knn16_bounds_bytes(N, q_count), dtype=torch.uint8, device=dev)
                    check((lib.meld_knn16_bounds_from_spheres if spheres_shared else lib.meld_knn16_bounds)(ptr(X_search), N, d, ptr(mean_search), ptr(scale_info), ptr(nmax), ptr(Rt), q_begin, q_count, ptr(seeds) if seeded_bounds else None, ptr(Qn) if seeded_bounds else None, nprod, ptr(tmpb), ptr(lb2), st), "meld_knn16_bounds")
                if direct:
                    pass
                elif want_lists:
                    # the tiles a block can rule out at its start thresholds, written down once (the count is the block's work)
                    step_list = torch.empty(n_blocks * n_tiles, dtype=torch.int32, device=dev)
                    step_cnt = torch.empty(n_blocks, dtype=torch.int32, device=dev)
                    check(lib.meld_knn16_step_lists(ptr(lb2), ptr(seeds), N, d, q_count, nprod, ptr(nmax), ptr(scale_info), 0 if cross else q_begin,
                                                    ptr(step_list), n_tiles, ptr(step_cnt), st), "meld_knn16_step_lists")
                    work = step_cnt
                elif self.block_order and n_blocks > 1:
                    # longest query blocks first (the dispatch follows the block index): see meld_knn16_block_work
                    work = torch.empty(n_blocks, dtype=torch.int32, device=dev)
                    check(lib.meld_knn16_block_work(ptr(lb2), ptr(seeds), N, d, q_count, nprod, ptr(nmax), ptr(scale_info), ptr(work), st), "meld_knn16_block_work")
                else:
                    work = None
                if self.block_order and work is not None and n_blocks > 1:
                    block_order = torch.argsort(work, descending=True, stable=True).to(torch.int32)
                tm.stop("bounds")
            # Few query blocks (a row shard, a mid-sized data set): with pruning the work of a block varies 12-fold and a
            # launch that fills the chip less than twice over ends when its heaviest block does (a 1/8 shard of 1M cells:
            # 7-10 ms instead of 26 / 8).  The references are then cut into slices -- blocks x slices workgroups, each
            # with its own candidate rows, merged afterwards -- so that the heavy blocks are shared out.
            main_slices = 1
            if will_prune and cand_thr is not None:
                resident = lib.meld_knn16_resident_blocks(d, nprod)
                if opt("MELD_KNN_MAIN_SLICES"):
                    main_slices = int(opt("MELD_KNN_MAIN_SLICES"))
                elif resident > 0 and n_blocks < 2 * resident:
                    main_slices = int(max(1, min(4, lib.meld_knn16_max_slices(ksel), -(-2 * resident // n_blocks), n_tiles // 64)))  # (more slices cost more in merging than they balance)
            # The partial-distance test of the principal frame as a pass of its own (meld_knn16_partial_filter): every listed (wave,
            # tile) pair is tested on K block 0 against the row's start threshold and the lists are thinned in place; the search
            # then stages whole tiles for the quarter of the pairs that survive (it keeps its own test per block of 32 references:
            # 30 % of the blocks of a surviving pair still stop behind K block 0).  MELD_KNN_TWO_PHASE=0: the round-5 form, every
            # listed tile staged in full by the one kernel that tests and searches.
            # (few query blocks -- a row shard, a mid-sized data set: main_slices > 1 -- keep the one-kernel form: a workgroup of the
            # filter pass walks its block's whole list, and a launch that fills the chip less than twice over ends when its longest
            # list does -- a 1/8 shard of 1M cells: 2.4-3.0 ms for the filter alone against 3.0 ms for the sliced search)
            two_phase = (step_list is not None and X_search is not X and seeds is not None
                         and (main_slices == 1 or opt("MELD_KNN_TWO_PHASE") == "2")
                         and opt("MELD_KNN_TWO_PHASE", "1") != "0" and opt("MELD_KNN16_EE") is None)
            partial_in_search = int(X_search is not X)
            tiles_b = tiles_done
            if two_phase:
                with _EventSpan("knn_filter", N=N, d=d, q=q_count):
                    check(lib.meld_knn16_partial_filter(ptr(Q), ptr(Qn), ptr(Rt), ptr(scale_info), ptr(nmax), d, q_count, ptr(seeds), ptr(step_list),
                                                        ptr(step_cnt), n_tiles, ptr(step_cnt), ptr(tiles_done), ptr(block_order), st), "meld_knn16_partial_filter")
                    if block_order is not None:  # (longest blocks first, by what is left of them)
                        block_order = torch.argsort(step_cnt, descending=True, stable=True).to(torch.int32)
                tiles_b = tiles_done[1:]  # (the search counts the pairs it computes, and its blocks, behind the filter's)
                if opt("MELD_KNN_LIST_STATS"):  # (development: how long the thinned lists are -- the longest one bounds the search from below)
                    sc = step_cnt.to(torch.float64)
                    print("[lists behind the filter] blocks %d  entries: mean %.0f  median %.0f  p90 %.0f  p99 %.0f  max %.0f" % (
                        sc.numel(), sc.mean(), sc.median(), torch.quantile(sc, 0.9), torch.quantile(sc, 0.99), sc.max()), file=sys.stderr)
                tm.stop("knn_filter")
            with _EventSpan("knn_topk", N=N, d=d, q=q_count):
                if main_slices > 1:
                    s_idx = torch.empty(main_slices * q_pad * cap, dtype=torch.int32, device=dev)
                    s_d2 = torch.empty(main_slices * q_pad * cap, dtype=torch.float32, device=dev)
                    s_cnt = torch.empty(main_slices * q_pad, dtype=torch.int32, device=dev)
                    s_thr = torch.full((main_slices, q_pad), float("inf"), dtype=torch.float32, device=dev)
                    if step_list is not None:
                        check(lib.meld_knn16_topk_listed_partial(ptr(Q), ptr(Qn), ptr(Rt), ptr(scale_info), NR, d, q_count, ksel, ptr(step_list), ptr(step_cnt), n_tiles, ptr(nmax), 0 if cross else q_begin, ptr(seeds), knn_cut, rfac, ptr(s_idx), ptr(s_d2), ptr(s_cnt), ptr(s_thr), ptr(tiles_b), ptr(block_order), main_slices, partial_in_search, st), "meld_knn16_topk_listed(sliced)")
                    else:
                        check(lib.meld_knn16_topk(ptr(Q), ptr(Qn), ptr(Rt), ptr(scale_info), NR, d, q_count, ksel, nprod, main_slices, ptr(lb2), ptr(nmax), 0 if cross else q_begin, ptr(seeds), knn_cut, rfac, ptr(s_idx), ptr(s_d2), ptr(s_cnt), ptr(s_thr), ptr(tiles_done), ptr(block_order), st), "meld_knn16_topk(sliced)")
                    check(lib.meld_knn16_merge_slices(ptr(s_idx), ptr(s_d2), ptr(s_cnt), q_count, ksel, main_slices, ptr(cand_idx), ptr(cand_d2), ptr(cand_cnt), st), "meld_knn16_merge_slices")
                    cand_thr.copy_(s_thr.amin(0))  # the merged row holds every reference below the smallest slice threshold
                    del s_idx, s_d2, s_cnt, s_thr
                elif step_list is not None:
                    check(lib.meld_knn16_topk_listed_partial(ptr(Q), ptr(Qn), ptr(Rt), ptr(scale_info), NR, d, q_count, ksel, ptr(step_list), ptr(step_cnt), n_tiles, ptr(nmax), 0 if cross else q_begin, ptr(seeds), knn_cut, rfac, ptr(cand_idx), ptr(cand_d2), ptr(cand_cnt), ptr(cand_thr), ptr(tiles_b), ptr(block_order), 1, partial_in_search, st), "meld_knn16_topk_listed")
                else:
                    check(lib.meld_knn16_topk(ptr(Q), ptr(Qn), ptr(Rt), ptr(scale_info), NR, d, q_count, ksel, nprod, 1, ptr(lb2), ptr(nmax), 0 if cross else q_begin, ptr(seeds), knn_cut, rfac, ptr(cand_idx), ptr(cand_d2), ptr(cand_cnt), ptr(cand_thr), ptr(tiles_done), ptr(block_order), st), "meld_knn16_topk")
                # the search is the one long launch of the build (26 of 45 ms at 1M cells) and the host has nothing to do
                # until its results are refined: work that does not depend on the graph (fit_transform's label
                # factorisation: a host-blocking copy + a few small launches on a side stream) is started here
                while _WHILE_SEARCHING:
                    _WHILE_SEARCHING.pop()()
            if opt("MELD_KNN_SAVE_THR") and cand_thr is not None:
                torch.save((cand_thr * scale_info[0] ** 2 * 1.0001).cpu(), opt("MELD_KNN_SAVE_THR"))
            used_prune, used_seed = lb2 is not None or step_list is not None, seeds is not None
            used_seeded_bounds = bool(will_prune and seeds is not None and self.seeded_bounds)
            used_block_order = block_order is not None
            used_step_lists = step_list is not None
            used_two_phase = bool(two_phase)
            del lb2, step_list, step_cnt
            KP = 16 * KB
            research = dict(Rt=Rt, scale_info=scale_info, KB=KB, BQ=BQ) if nprod == 1 else None
        else:
            research = None
            # fp32 operands on v_mfma_f32_32x32x2_f32 (knn.hip)
            KP = lib.meld_knn_padded_dim(d)
            if KP < 0:
                check(KP, "meld_knn_padded_dim")
            TS, BQ = lib.meld_knn_tile_refs(), lib.meld_knn_block_queries()
            cap = lib.meld_knn_row_capacity(ksel)
            if cap < 0:
                check(cap, "meld_knn_row_capacity")
            err_coef = lib.meld_knn_error_coef(d)
            err_lin = 0.0
            n_tiles = (N + TS - 1) // TS
            Rt = torch.empty(n_tiles * KP * TS, dtype=torch.float32, device=dev)
            check(lib.meld_knn_prepare_refs(ptr(X), N, d, ptr(mean), KP, ptr(Rt), ptr(norm2), ptr(nmax), st), "meld_knn_prepare_refs")
            q_pad = ((q_count + BQ - 1) // BQ) * BQ
            Q = torch.empty(q_pad * KP, dtype=torch.float32, device=dev)
            check(lib.meld_knn_prepare_queries(ptr(X), N, d, ptr(mean), KP, q_begin, q_count, ptr(Q), st), "meld_knn_prepare_queries")
            tm.stop("prepare")
            cand_idx = torch.empty(q_pad * cap, dtype=torch.int32, device=dev)
            cand_d2 = torch.empty(q_pad * cap, dtype=torch.float32, device=dev)
            cand_cnt = torch.empty(q_pad, dtype=torch.int32, device=dev)
            with _EventSpan("knn_topk", N=N, d=d, q=q_count):
                check(lib.meld_knn_topk(ptr(Q), ptr(Rt), N, KP, q_count, ksel, ptr(cand_idx), ptr(cand_d2), ptr(cand_cnt), st), "meld_knn_topk")
        tm.stop("knn_topk")
        del Q
        if research is None:
            del Rt

        # exact refinement + alpha-decay kernel
        max_rank = 0 if knn_max is None else int(knn_max) + 1  # (self counted, as graphtools counts it)
        bw = torch.empty(q_count, dtype=torch.float64, device=dev)
        cand_val = torch.empty(q_count * ksel, dtype=torch.float64, device=dev)
        keep_cnt = torch.empty(q_count, dtype=torch.int32, device=dev)
        flag_rows = torch.empty(q_count, dtype=torch.int32, device=dev)
        n_flag = torch.zeros(1, dtype=torch.int32, device=dev)
        nmax_used = nmax
        if force_fallback:  # test hook: an infinite error bound flags every row
            nmax_used = torch.full((1,), float("inf"), dtype=torch.float32, device=dev)
        if opt("MELD_REFINE_STATS") and bw_fixed is None:  # (development: how many candidate rows the refinement gathers per row)
            c2 = cand_d2.view(-1, cap)[:q_count].to(torch.float64)
            cn = cand_cnt[:q_count].clamp(max=ksel)
            E_ = float(err_coef) * nmax.to(torch.float64) + float(err_lin) * torch.sqrt(norm2[q_begin : q_begin + q_count].to(torch.float64) * nmax.to(torch.float64))
            rf_ = max((1.0 if math.isinf(decay) else float((-math.log(thresh)) ** (1.0 / decay))) * float(bw_scale), 1.0)
            skip_ = rf_ * rf_ * (c2[:, min(knn, cap - 1)] + E_) + E_
            g_ = ((c2 <= skip_[:, None]) & (torch.arange(cap, device=dev)[None, :] < cn[:, None])).sum(1).to(torch.float64)
            print("[refine] rows %d  listed: mean %.1f  gathered: mean %.1f  median %.0f  p90 %.0f  p99 %.0f  max %.0f" % (
                q_count, cn.to(torch.float64).mean(), g_.mean(), g_.median(), torch.quantile(g_, 0.9), torch.quantile(g_, 0.99), g_.max()), file=sys.stderr)
        check(
            lib.meld_knn_refine(
                ptr(X), N, d, q_begin, q_count, ptr(cand_idx), ptr(cand_d2), ptr(cand_cnt), ptr(cand_thr), ksel, cap, knn, float(decay),
                float(thresh), ptr(nmax_used), float(err_coef), ptr(norm2), float(err_lin), ptr(bw), ptr(cand_val), ptr(keep_cnt),
                ptr(flag_rows), ptr(n_flag), None, 0, None, float(bw_scale), ptr(bw_fixed), max_rank, st,
            ),
            "meld_knn_refine",
        )
        n_flag_h = int(n_flag.item())
        n_flag_stage1 = n_flag_h
        tm.stop("refine")

        # second search stage: rows the reduced-precision pass could not certify are searched again with
        # the full hi/lo split (a few % of the rows); only what that cannot certify either goes to the
        # exact sweep
        if research is not None and n_flag_h > 0 and not force_fallback:
            rows2 = torch.sort(flag_rows[:n_flag_h]).values.contiguous()
            KB, BQ2 = research["KB"], research["BQ"]
            q2_pad = ((n_flag_h + BQ2 - 1) // BQ2) * BQ2
            Q2 = torch.empty(q2_pad * lib.meld_knn16_query_bytes(d), dtype=torch.uint8, device=dev)
            Qn2 = torch.empty(q2_pad, dtype=torch.float32, device=dev)
            check(lib.meld_knn16_prepare_rows(ptr(X_search), N, d, ptr(mean_search), ptr(research["scale_info"]), q_begin, ptr(rows2), n_flag_h, ptr(Q2), ptr(Qn2), st), "meld_knn16_prepare_rows")
            # few queries: cut the references into slices so that the re-search fills the chip
            n_blocks2 = q2_pad // BQ2
            resident = lib.meld_knn16_resident_blocks(d, 3)
            if resident < 0:
                check(resident, "meld_knn16_resident_blocks")
            n_slices = int(max(1, min(lib.meld_knn16_max_slices(ksel), 2 * resident // max(n_blocks2, 1), n_tiles)))
            # Start the thresholds of the re-search at a bound instead of +inf: the first pass found ksel
            # references with approximate d2 <= tau, so the true ksel-th distance is <= tau + E1 and its
            # full-precision approximation <= tau + E1 + E3 -- nothing above that can enter the list.
            # (Without it every slice selects from scratch: 19k appends per query at 1M cells.)
            thr2 = torch.empty(q2_pad, dtype=torch.float32, device=dev)
            check(lib.meld_knn16_research_thresholds(ptr(rows2), n_flag_h, q_begin, ptr(cand_cnt), ptr(cand_d2), cap, ksel, ptr(norm2), ptr(nmax),
                                                     float(err_coef), float(err_lin), float(lib.meld_knn16_error_coef(3, d)),
                                                     ptr(research["scale_info"]), ptr(thr2), st), "meld_knn16_research_thresholds")
            c2_idx = torch.empty(n_slices * q2_pad * cap, dtype=torch.int32, device=dev)
            c2_d2 = torch.empty(n_slices * q2_pad * cap, dtype=torch.float32, device=dev)
            c2_cnt = torch.empty(n_slices * q2_pad, dtype=torch.int32, device=dev)
            with _EventSpan("knn_topk_stage2", N=N, d=d, q=n_flag_h):
                check(lib.meld_knn16_topk(ptr(Q2), ptr(Qn2), ptr(research["Rt"]), ptr(research["scale_info"]), NR, d, n_flag_h, ksel, 3, n_slices, None, ptr(nmax), 0, ptr(thr2), 0, 1.0, ptr(c2_idx), ptr(c2_d2), ptr(c2_cnt), None, None, None, st), "meld_knn16_topk(stage 2)")
                if n_slices > 1:
                    m_idx = torch.empty(q2_pad * cap, dtype=torch.int32, device=dev)
                    m_d2 = torch.empty(q2_pad * cap, dtype=torch.float32, device=dev)
                    m_cnt = torch.empty(q2_pad, dtype=torch.int32, device=dev)
                    check(lib.meld_knn16_merge_slices(ptr(c2_idx), ptr(c2_d2), ptr(c2_cnt), n_flag_h, ksel, n_slices, ptr(m_idx), ptr(m_d2), ptr(m_cnt), st), "meld_knn16_merge_slices")
                    c2_idx, c2_d2, c2_cnt = m_idx, m_d2, m_cnt
            n_flag.zero_()
            check(
                lib.meld_knn_refine(
                    ptr(X), N, d, q_begin, n_flag_h, ptr(c2_idx), ptr(c2_d2), ptr(c2_cnt), None, ksel, cap, knn, float(decay),
                    float(thresh), ptr(nmax), float(lib.meld_knn16_error_coef(3, d)), None, 0.0, ptr(bw), ptr(cand_val), ptr(keep_cnt),
                    ptr(flag_rows), ptr(n_flag), ptr(rows2), cap, ptr(cand_idx), float(bw_scale), ptr(bw_fixed), max_rank, st,
                ),
                "meld_knn_refine(stage 2)",
            )
            del Q2, c2_idx, c2_d2, c2_cnt
            n_flag_stale = True
            tm.stop("knn_stage2")
        else:
            n_flag_stale = False
        research = None
        Rt = None
        keep_off = _scan_i32(lib, keep_cnt, st)
        # ONE read-back for the three scalars the host wants here (each one is an idle gap of the GPU of ~50 us: nothing is queued
        # behind it): rows still flagged after the second stage, kept entries, (wave, tile) pairs the first pass computed
        heads = [n_flag[0].to(torch.int64), keep_off[q_count]] + ([tiles_done[0], tiles_done[2] if used_two_phase else tiles_done[1], tiles_done[1]] if tiles_done is not None else [])
        heads_h = torch.stack(heads).tolist()
        if n_flag_stale:
            n_flag_h = int(heads_h[0])
        m_main = int(heads_h[1])
        tiles_done_h = int(heads_h[2]) if tiles_done is not None else None
        blocks_on_h = int(heads_h[3]) if tiles_done is not None and X_search is not X else None
        pairs_kept_h = int(heads_h[4]) if tiles_done is not None and used_two_phase else None  # (wave, tile) pairs the filter pass left to the search
        if pairs_kept_h is not None and lib.meld_knn16_kblocks(d) > 7:
            blocks_on_h = 2 * pairs_kept_h  # (beyond seven K blocks the search behind the filter has no test of its own: both blocks of every pair it is handed)

        # Many uncertified rows with a short candidate list (dense low-dimensional data: more than ksel cells
        # inside the radius inflated by the search-error allowance): search once more with the longest list
        # instead of sweeping them one by one (1M cells in the plane, knn = 15: 292k rows through the sweep at
        # ksel = 64, 0.63 s; none at ksel = 128, 32 ms).  Same graph either way.
        if (n_flag_h > max(1024, q_count // 100) and ksel < 128 and search == "f16x3" and not force_fallback
                and opt("MELD_KNN_RETRY", "1") != "0"):
            # (comm is NOT forwarded on purpose: only the ranks that need the retry take it, so it must not issue collectives
            # -- the shared-spheres all-gather of the first try is skipped, every rank computes all spheres itself)
            out = self.directed_kernel_coo(X, q_begin, q_count, knn, decay, thresh, 128, tm=tm, force_fallback=False, n_refs=n_refs,
                                           assemble=assemble, bw_scale=bw_scale, bw_fixed=bw_fixed, col_stats=col_stats, knn_max=knn_max, symm=symm,
                                           count_rows_ge=count_rows_ge, frame_axes=frame_axes)
            out[3]["ksel_retry_from"] = int(ksel)
            out[3]["n_flagged_rows_first_try"] = int(n_flag_h)
            return out

        # exact sweep for rows the candidate list could not certify
        knn_chk = knn if bw_fixed is None else 2**31 - 1  # (a given bandwidth is not verified against the neighbour count)
        n_rebandwidth = 0
        fb_total = 0
        fb_off = fb_col = fb_val = None
        if n_flag_h > 0:
            flag_rows = torch.sort(flag_rows[:n_flag_h]).values.contiguous()  # deterministic order
            fb_cnt = torch.empty(n_flag_h, dtype=torch.int32, device=dev)
            err = torch.zeros(1, dtype=torch.int32, device=dev)
            cursor = torch.zeros(n_flag_h, dtype=torch.int32, device=dev)  # count pass: references closer than bw
            check(
                lib.meld_knn_radius_exact(
                    ptr(X), NR, d, q_begin, ptr(flag_rows), n_flag_h, ptr(bw), knn_chk, float(decay), float(thresh), 0,
                    ptr(fb_cnt), None, ptr(cursor), None, None, ptr(err), float(bw_scale), st,
                ),
                "meld_knn_radius_exact(count)",
            )
            if int(err.item()) != 0:
                # rows whose candidate list missed one of their knn nearest cells (marked fb_cnt = -1: more than knn
                # references are strictly closer than the bandwidth the list implied): their bandwidth is recomputed
                # exactly over all references -- a rare library path (fp64 screen + direct differences) -- and the
                # sweep is counted again (graphtools re-searches such rows with more neighbours)
                bad = torch.nonzero(fb_cnt < 0).reshape(-1)
                rows_bad = flag_rows[bad].to(torch.int64)
                bw[rows_bad] = _exact_bandwidth(X, q_begin + rows_bad, knn, n_refs=NR)
                fb_cnt.zero_()
                cursor.zero_()
                err.zero_()
                check(
                    lib.meld_knn_radius_exact(
                        ptr(X), NR, d, q_begin, ptr(flag_rows), n_flag_h, ptr(bw), knn_chk, float(decay), float(thresh), 0,
                        ptr(fb_cnt), None, ptr(cursor), None, None, ptr(err), float(bw_scale), st,
                    ),
                    "meld_knn_radius_exact(recount)",
                )
                if int(err.item()) != 0:
                    raise NotImplementedError(
                        "degenerate neighbourhoods: the exact sweep could not settle the bandwidth of {} rows".format(int((fb_cnt < 0).sum())))
                n_rebandwidth = int(bad.shape[0])
            fb_off = _scan_i32(lib, fb_cnt, st)
            fb_total = int(fb_off[n_flag_h].item())
            # (a radius that covers most of the data -- small decay with a small thresh -- makes the graph effectively dense: say so
            # instead of failing inside an allocation; the reference's scipy matrices would be as large)
            need = 12 * fb_total + 32 * (m_main + fb_total)
            if need > torch.cuda.get_device_properties(dev).total_memory:
                raise MemoryError(
                    "the kernel radius covers {:.3g} neighbours per cell on average: the graph would hold {:.3g} entries ({:.0f} GB to "
                    "assemble) -- raise decay or thresh".format((m_main + fb_total) / max(q_count, 1), float(m_main + fb_total), need / 1e9))
            fb_col = torch.empty(max(fb_total, 1), dtype=torch.int32, device=dev)
            fb_val = torch.empty(max(fb_total, 1), dtype=torch.float64, device=dev)
            check(  # (the count pass left the cursors at zero)
                lib.meld_knn_radius_exact(
                    ptr(X), NR, d, q_begin, ptr(flag_rows), n_flag_h, ptr(bw), knn_chk, float(decay), float(thresh), 1,
                    None, ptr(fb_off), ptr(cursor), ptr(fb_col), ptr(fb_val), None, float(bw_scale), st,
                ),
                "meld_knn_radius_exact(fill)",
            )
            if knn_max is not None and fb_total > 0 and int(fb_cnt.max()) > int(knn_max):
                # knn_max on the rows the sweep filled (everything inside the radius, unranked): keep the knn_max largest kernel
                # values of a row (= its nearest cells; ties by column) -- a rare path, a few library sorts over the swept entries
                rows_e = torch.repeat_interleave(torch.arange(n_flag_h, device=dev), fb_cnt.to(torch.int64))
                o = torch.argsort(fb_col[:fb_total].to(torch.int64), stable=True)
                o = o[torch.argsort(-fb_val[:fb_total][o], stable=True)]
                o = o[torch.argsort(rows_e[o], stable=True)]
                rank = torch.arange(fb_total, device=dev) - fb_off[:n_flag_h][rows_e[o]]
                keep = o[rank < int(knn_max)]
                keep = keep[torch.argsort(rows_e[keep], stable=True)]
                fb_col, fb_val = fb_col[keep].contiguous(), fb_val[keep].contiguous()
                fb_cnt = torch.clamp(fb_cnt, max=int(knn_max))
                fb_off = _scan_i32(lib, fb_cnt, st)
                fb_total = int(fb_off[n_flag_h].item())
        tm.stop("radius_exact")
        rows_at_least = None
        if count_rows_ge is not None:
            tot = keep_cnt.to(torch.int64)
            if n_flag_h > 0 and fb_total > 0:
                tot = tot.clone()
                tot[flag_rows[:n_flag_h].to(torch.int64)] = fb_cnt.to(torch.int64)
            rows_at_least = int(((tot + 1) >= int(count_rows_ge)).sum())  # (+ 1: the cell itself, K_ii = 1 is carried analytically)

        M = m_main + fb_total
        assembled = None
        if assemble and M > 0 and q_begin == 0 and q_count == NR and not cross and opt("MELD_ASSEMBLE", "bucket") == "bucket" \
                and opt("MELD_ASSEMBLE_FUSED", "1") != "0":
            # single GPU, every row local: the kept candidates go straight into the row buckets of the symmetrisation
            # (meld_coo_emit_scatter) instead of through 2 M (key, value) pairs -- 512 MB written and read back at 1M cells
            B = int(lib.meld_csr_bucket_slots())
            if ksel <= B and q_count * B * 12 <= torch.cuda.mem_get_info(dev)[0] // 4:
                cursor = keep_cnt.clone()
                if n_flag_h > 0 and fb_total > 0:
                    cursor[flag_rows[:n_flag_h].to(torch.int64)] = fb_cnt
                tcol = torch.empty(q_count * B, dtype=torch.int32, device=dev)
                tval = torch.empty(q_count * B, dtype=torch.float64, device=dev)
                check(lib.meld_coo_emit_scatter(q_count, ptr(cand_idx), ptr(cand_val), ksel, cap, ptr(keep_cnt), ptr(flag_rows), n_flag_h,
                                                ptr(fb_off), ptr(fb_col), ptr(fb_val), fb_total, ptr(cursor), ptr(tcol), ptr(tval), st),
                      "meld_coo_emit_scatter")
                tm.stop("coo_emit")
                assembled = self._finish_buckets(cursor, tcol, tval, q_count, sums_diag=1.0, symm=symm)  # None: a bucket overflowed / a column thrice
                if assembled is not None and self.last_row_sums is not None:
                    assembled = assembled + (self.last_row_sums[1],)  # (kernel row sums incl. the unit diagonal)
                tm.stop("symmetrize")
                del cursor, tcol, tval
        keys = vals = None
        if assembled is None:
            keys = torch.empty(2 * M, dtype=torch.int64, device=dev)
            vals = torch.empty(2 * M, dtype=torch.float64, device=dev)
        if M > 0 and assembled is None:
            check(
                lib.meld_coo_emit(
                    q_begin, q_count, ptr(cand_idx), ptr(cand_val), ptr(cand_cnt), ksel, cap, ptr(keep_off), ptr(flag_rows),
                    n_flag_h, ptr(fb_off), ptr(fb_col), ptr(fb_val), m_main, M, ptr(keys), ptr(vals), st,
                ),
                "meld_coo_emit",
            )
        tm.stop("coo_emit")
        nprod_used = nprod if search == "f16x3" else self.nprod
        info = dict(ksel=int(ksel), KP=int(KP), search=search, nprod=nprod_used, n_flagged_rows=n_flag_h,
                    # which of the search options (constructor arguments / MELD_KNN_* ablation switches) were in effect
                    prune=bool(used_prune), radius_cut=bool(cand_thr is not None), seed=bool(used_seed), seeded_bounds=bool(used_seeded_bounds), block_order=bool(used_block_order), step_lists=bool(used_step_lists), principal_frame=bool(X_search is not X), two_phase=bool(used_two_phase), seed_side=int(opt("MELD_KNN_SEED_SIDE", "0")),
                    n_rows_bandwidth_recomputed=n_rebandwidth,
                    n_researched_rows=n_flag_stage1 if search == 'f16x3' and nprod_used == 1 else 0, nnz_directed=M,
                    # (wave, tile) pairs the first search pass computed (all of them without pruning)
                    wave_tiles_done=tiles_done_h, blocks_past_partial_test=blocks_on_h, pairs_past_filter=pairs_kept_h)
        if rows_at_least is not None:
            info["rows_with_at_least"] = rows_at_least
        if assembled is not None:
            info["assembled"] = assembled  # (rowptr, col, val) of the symmetrised rows: the caller skips assemble_rows
        return keys, vals, bw, info

    # ---- A4: (K + K^T)/2 rows [row_begin, row_begin + n_rows) from unsorted COO ---------------------
    def sort_pairs(self, keys, vals, N):
        lib, st = self.lib, _stream()
        n = int(keys.shape[0])
        keys2 = torch.empty_like(keys)
        vals2 = torch.empty_like(vals)
        if n == 0:
            return keys2, vals2
        tb = lib.meld_sort_temp_bytes(n)
        tmp = torch.empty(tb, dtype=torch.uint8, device=keys.device)
        end_bit = 32 + max(1, int(N - 1).bit_length())
        check(lib.meld_sort_pairs_u64_f64(ptr(keys), ptr(keys2), ptr(vals), ptr(vals2), n, end_bit, ptr(tmp), tb, st), "meld_sort_pairs_u64_f64")
        return keys2, vals2

    def partition_remote(self, keys, vals, rows_per_rank, world, rank, cap):
        """Row-sharded build: the entries owed to the other ranks in a fixed-capacity send buffer [world, 2, cap] (keys |
        value bits, sentinel key ~0 in unused slots) and the per-owner counts, all on the device
        (``meld_coo_partition_remote``)."""
        dev = keys.device
        counts = torch.empty(world, dtype=torch.int32, device=dev)
        send = torch.empty(world * 2 * cap, dtype=torch.int64, device=dev)
        check(self.lib.meld_coo_partition_remote(ptr(keys), ptr(vals), int(keys.shape[0]), int(rows_per_rank), int(world), int(rank),
                                                 int(cap), ptr(counts), ptr(send), _stream()), "meld_coo_partition_remote")
        return send, counts

    def _finish_buckets(self, cursor, tcol, tval, n_rows, sums_diag=None, symm=(0, 0.0)):
        """Row buckets (meld_coo_scatter_rows / meld_coo_emit_scatter) -> CSR: every bucket sorted by column and its pairs
        of equal columns summed inside one wave, then compacted.  None when a bucket overflowed or a column occurs more
        than twice (the caller takes the sort-based path, whose summation order is defined)."""
        lib, st, dev = self.lib, _stream(), cursor.device
        i32 = dict(dtype=torch.int32, device=dev)
        ucnt = torch.empty(n_rows, **i32)
        flags = torch.empty(1, **i32)
        check(lib.meld_csr_rows_sort_merge(ptr(cursor), n_rows, ptr(tcol), ptr(tval), ptr(ucnt), ptr(flags), int(symm[0]), float(symm[1]), st), "meld_csr_rows_sort_merge")
        rowptr = _scan_i32(lib, ucnt, st)
        nnz, flag = (int(v) for v in torch.stack([rowptr[n_rows], flags[0].to(torch.int64)]).tolist())  # (one read-back)
        if flag != 0:
            return None
        col = torch.empty(nnz, **i32)
        val = torch.empty(nnz, dtype=torch.float64, device=dev)
        self.last_row_sums = None
        if nnz > 0 and sums_diag is not None:
            # (the rows' sums on the way out of the buckets: meld_csr_row_sums' bits without its pass over the values)
            sums = torch.empty(n_rows, dtype=torch.float64, device=dev)
            check(lib.meld_csr_compact_rows_sums(ptr(rowptr), n_rows, ptr(tcol), ptr(tval), ptr(col), ptr(val), float(sums_diag), ptr(sums), st),
                  "meld_csr_compact_rows_sums")
            self.last_row_sums = (sums_diag, sums)
        elif nnz > 0:
            check(lib.meld_csr_compact_rows(ptr(rowptr), n_rows, ptr(tcol), ptr(tval), ptr(col), ptr(val), st), "meld_csr_compact_rows")
        self.last_assemble = "bucket"
        return rowptr, col, val

    def assemble_rows(self, keys, vals, row_begin, n_rows, N, foreign=False, symm=(0, 0.0)):
        """Sum duplicate keys, build the CSR (sorted rows) of the local rows: by row buckets sorted inside one wave
        each (include/meld_hip.h, meld_coo_row_counts ...), or -- for the inputs that path refuses, and with
        ``MELD_ASSEMBLE=sort`` -- by a global radix sort + reduce-by-key.  ``foreign``: the input may hold entries of
        rows outside the slice (the fixed-capacity exchange of the sharded build: other ranks' rows, sentinel keys);
        the bucket path ignores them, the sort path drops them first."""
        lib, st, dev = self.lib, _stream(), keys.device
        n = int(keys.shape[0])
        B = int(lib.meld_csr_bucket_slots())
        # the buckets cost 12 B x B slots per row whatever the degree (3 KB per row: 3 GB at 1M rows): above a quarter of
        # the free memory -- very large or tightly sharded runs -- the sort path (32 B per entry) is taken instead
        bucket_bytes = n_rows * B * 12
        fits = bucket_bytes <= torch.cuda.mem_get_info(dev)[0] // 4 if n_rows > 0 else True
        if n > 0 and n_rows > 0 and fits and opt("MELD_ASSEMBLE", "bucket") != "sort":
            i32 = dict(dtype=torch.int32, device=dev)
            cursor = torch.empty(n_rows, **i32)
            tcol = torch.empty(n_rows * B, **i32)
            tval = torch.empty(n_rows * B, dtype=torch.float64, device=dev)
            check(lib.meld_coo_scatter_rows(ptr(keys), ptr(vals), n, row_begin, n_rows, ptr(cursor), ptr(tcol), ptr(tval), st), "meld_coo_scatter_rows")
            done = self._finish_buckets(cursor, tcol, tval, n_rows, symm=symm)
            if done is not None:
                return done
            del cursor, tcol, tval
        self.last_assemble = "sort" if n > 0 else "empty"
        if foreign and n > 0:
            rows = keys >> 32  # (the sentinel ~0 is -1 as int64: its row is negative)
            own = (rows >= row_begin) & (rows < row_begin + n_rows)
            keys, vals = keys[own].contiguous(), vals[own].contiguous()
            n = int(keys.shape[0])
        keys2, vals2 = self.sort_pairs(keys, vals, N)
        if symm[0] != 0 and n > 0:
            # kernel_symm "*" / "mnn" behind the global sort (overfull row buckets -- hub cells --, MELD_ASSEMBLE=sort): the two HALVES
            # of a key sit side by side; the rules of meld_csr_rows_sort_merge (csrc/assemble.hip) on them, as library segment
            # reductions -- a rare path.  Symmetric functions of the pair: (i, j) and (j, i) get the same bits.
            first = torch.ones(n, dtype=torch.bool, device=dev)
            first[1:] = keys2[1:] != keys2[:-1]
            seg = torch.cumsum(first, 0) - 1
            nseg = int(seg[-1].item()) + 1
            cnt = torch.bincount(seg, minlength=nseg)
            if int(cnt.max().item()) > 2:
                raise NotImplementedError("kernel_symm other than '+': an entry of the kernel occurs more than once per direction")
            hi = torch.zeros(nseg, dtype=torch.float64, device=dev).scatter_reduce_(0, seg, vals2, "amax", include_self=False)
            lo = torch.zeros(nseg, dtype=torch.float64, device=dev).scatter_reduce_(0, seg, vals2, "amin", include_self=False)
            lo = torch.where(cnt == 2, lo, torch.zeros_like(lo))  # (the missing direction counts as 0)
            if symm[0] == 1:
                v, keep = 4.0 * hi * lo, cnt == 2
            else:
                v = 2.0 * (float(symm[1]) * lo + (1.0 - float(symm[1])) * hi)
                keep = v != 0.0
            ukeys, uvals = keys2[first][keep].contiguous(), v[keep].contiguous()
            nnz = int(ukeys.shape[0])
        else:
            tb = lib.meld_merge_temp_bytes(max(n, 1))
            tmp = torch.empty(tb, dtype=torch.uint8, device=dev)
            n_unique = torch.zeros(1, dtype=torch.int64, device=dev)
            ukeys = torch.empty_like(keys2)
            uvals = torch.empty_like(vals2)
            check(lib.meld_coo_merge(ptr(keys2), ptr(vals2), n, ptr(ukeys), ptr(uvals), ptr(n_unique), ptr(tmp), tb, st), "meld_coo_merge")
            nnz = int(n_unique.item())
        rowptr = torch.empty(n_rows + 1, dtype=torch.int64, device=dev)
        col = torch.empty(nnz, dtype=torch.int32, device=dev)
        check(lib.meld_csr_from_keys(ptr(ukeys), nnz, row_begin, n_rows, ptr(rowptr), ptr(col), st), "meld_csr_from_keys")
        return rowptr, col, uvals[:nnz].clone()

    def gather_rows(self, X, perm):
        """X[perm] for an fp64 [N, d] matrix (``meld_gather_rows_f64``)."""
        X = X.contiguous()
        perm = perm.to(torch.int64).contiguous()
        out = torch.empty((int(perm.shape[0]), int(X.shape[1])), dtype=torch.float64, device=X.device)
        check(self.lib.meld_gather_rows_f64(ptr(X), ptr(perm), int(perm.shape[0]), int(X.shape[1]), ptr(out), _stream()), "meld_gather_rows_f64")
        return out

    def row_sums(self, rowptr, val, n_rows, diag):
        out = torch.empty(n_rows, dtype=torch.float64, device=rowptr.device)
        check(self.lib.meld_csr_row_sums(ptr(rowptr), ptr(val), n_rows, float(diag), ptr(out), _stream()), "meld_csr_row_sums")
        return out

    def anisotropy(self, rowptr, col, val, n_rows, ksum_all, row_off, a):
        check(self.lib.meld_csr_anisotropy(ptr(rowptr), ptr(col), ptr(val), n_rows, ptr(ksum_all), row_off, float(a), _stream()), "meld_csr_anisotropy")

    def anisotropy_degrees(self, rowptr, col, val, n_rows, ksum_all, row_off, a):
        """``anisotropy`` and the row sums of its result in one pass (``meld_csr_anisotropy_degrees``: the same bits as
        ``row_sums(..., 0.0)`` afterwards)."""
        dw = torch.empty(n_rows, dtype=torch.float64, device=val.device)
        check(self.lib.meld_csr_anisotropy_degrees(ptr(rowptr), ptr(col), ptr(val), n_rows, ptr(ksum_all), row_off, float(a), ptr(dw), _stream()),
              "meld_csr_anisotropy_degrees")
        return dw

    # ---- A9 / A10: operator steps -----------------------------------------------------------------
    def dot_slots(self):
        return self.lib.meld_spmm_dot_slots()

    # ---- panel-tiled copy of W for the recurrence (csrc/spmm_tiled.hip) -----------------------------
    shards_spheres = True  # directed_kernel_coo(comm=...) splits the tile spheres of the pruning table over the ranks
    PT_MIN_ROWS = 65536  # below this the CSR-stream kernel is launch-bound anyway and the layout does not pay

    _pt_selfcheck = {}  # device index -> bool (once per device and process, i.e. per load of the library)

    @classmethod
    def _pt_self_check(cls):
        """Once per device and process, before the tiled recurrence kernel is trusted: steps on a small random symmetric
        matrix through both kernels, for every instantiation the product launches -- p = 2 and p = 1 on fp64 values (a 3-column
        step is one launch of each) and the p = 1 kernel on the fp32 copy of the values (the SpMV of the lmax estimate).  The
        tiled kernel keeps loads in flight in registers it names itself and counts its waits by hand; a toolchain that broke
        those assumptions would return stale data, not an error -- a mismatch here keeps every graph on the CSR-stream kernel
        (and says so).  (The assembly-level guard of the same assumption runs at build time: ``meld_amd.build``.)"""
        dev_i = int(torch.cuda.current_device())
        if dev_i in cls._pt_selfcheck:
            return cls._pt_selfcheck[dev_i]
        cls._pt_selfcheck[dev_i] = ok = True
        from scipy import sparse

        rng = np.random.default_rng(0)
        n = 6000
        A = sparse.random(n, n, density=20.0 / n, random_state=0, format="csr", dtype=np.float64)
        A.data = rng.random(A.nnz) + 0.1
        W = ((A + A.T) * 0.5).tocsr()
        W.setdiag(0)
        W.eliminate_zeros()
        W.sort_indices()
        x3 = torch.from_numpy(rng.random((n, 3))).cuda()
        x1 = x3[:, 0].contiguous()
        outs = {}
        for mode in ("tiled", "csr"):
            G = DeviceGraph.from_scipy(W)
            ops = HipOps(spmm=mode)
            if mode == "tiled" and ops.pt_layout(G, _checking=True) is None:
                ok = False
                break
            y3 = torch.empty_like(x3)
            ops.cheby_step(G, 3, x3, 0, x3, y3, None, 0.7, -0.2, -1.0, 0.0)  # one launch of the p = 2 and one of the p = 1 kernel
            # the SpMV of the device-resident Lanczos loops: scalars from device memory, fp32 values on the tiled layout
            state = torch.zeros(8, dtype=torch.float64, device="cuda")
            state[3], state[4] = 0.9, -0.3
            dots = torch.zeros(2 * ops.dot_slots(), dtype=torch.float64, device="cuda")
            y1 = torch.empty_like(x1)
            ops.lanczos_spmv(G, x1, x1.clone(), y1, state, dots)
            outs[mode] = (y3, y1, dots[: ops.dot_slots()].sum())
        if ok:
            t, c = outs["tiled"], outs["csr"]
            e3 = float((t[0] - c[0]).abs().max() / c[0].abs().max())
            e1 = float((t[1] - c[1]).abs().max() / c[1].abs().max())
            ed = float((t[2] - c[2]).abs() / c[2].abs())
            ok = e3 < 1e-12 and e1 < 1e-6 and ed < 1e-6  # (fp32 values: 6e-8 per weight)
        cls._pt_selfcheck[dev_i] = ok
        if not ok:
            import warnings

            warnings.warn("meld_amd: the panel-tiled recurrence kernel failed its self-check against the CSR-stream kernel; "
                          "every graph stays on the CSR-stream kernel", RuntimeWarning)
        return ok

    def pt_layout(self, G, _checking=False):
        """The panel-tiled layout of ``G``'s local rows, built on first use and kept on the graph
        (``G.pt``); None when the graph stays on the CSR-stream kernel (small graphs, ``spmm="csr"``, or a
        graph the builder cannot lay out -- recorded in ``G.info["spmm"]``)."""
        pt = getattr(G, "pt", None)
        if pt is not None:
            return pt if pt is not False else None
        mode = getattr(self, "spmm", None) or opt("MELD_SPMM", "auto")
        G.info["spmm"] = "csr"
        G.pt = False
        if mode == "csr" or G.n_rows == 0 or G.nnz == 0 or not G.val.is_cuda:
            return None
        if mode == "auto" and G.n_rows < self.PT_MIN_ROWS:
            return None
        if not _checking and not self._pt_self_check():
            G.info["spmm"] = "csr (tiled kernel failed its self-check)"
            return None
        from ._lib import PtLayout
        import ctypes as C

        lib, dev = self.lib, G.val.device
        nb = int(lib.meld_pt_num_blocks(G.n_rows))
        slen = int(lib.meld_pt_stream_len(G.nnz, nb))
        i32 = dict(dtype=torch.int32, device=dev)
        t = dict(
            blk_row=torch.empty(nb + 1, **i32), blk_ntile=torch.empty(nb, **i32), blk_ndist=torch.empty(nb, **i32),
            seg=torch.empty(int(lib.meld_pt_seg_len(nb)), **i32), list_cols=torch.empty(G.nnz, **i32),
            pval=torch.empty(slen, dtype=torch.float64, device=dev), pidx=torch.empty(slen, **i32),
            pval32=torch.empty(slen, dtype=torch.float32, device=dev),  # fp32 values for the lmax estimate's SpMV
            cdesc=torch.empty(int(lib.meld_pt_desc_len(nb)), dtype=torch.int16, device=dev),
        )
        status = torch.zeros(1, **i32)
        lay = PtLayout(*(t[k].data_ptr() for k in ("blk_row", "blk_ntile", "blk_ndist", "seg", "list_cols", "pval", "pidx")), nb,
                       t["pval32"].data_ptr(), slen, t["cdesc"].data_ptr())
        codes = torch.empty(G.nnz, **i32)  # scratch of the builder
        # in-block pairs are stored once (W symmetric); the builder verifies the symmetry of every block's own
        # square and reports status 5 otherwise (a weight matrix uploaded from elsewhere): built again unfolded
        fold = opt("MELD_SPMM_FOLD", "1") != "0"
        for symmetric in ((1, 0) if fold else (0,)):
            with _EventSpan("pt_build", N=G.N, nnz=G.nnz):
                check(lib.meld_pt_build(ptr(G.rowptr), ptr(G.col), ptr(G.val), G.n_rows, G.n_pad, G.row_begin, symmetric,
                                        C.byref(lay), ptr(codes), ptr(status), _stream()), "meld_pt_build")
            st = int(status.item())
            if st != 5:
                break
        del codes
        G.info["spmm_fold"] = bool(symmetric) and st == 0
        if st != 0:  # cannot be laid out (see include/meld_hip.h): stay on the CSR-stream kernel
            G.info["spmm"] = "csr (tiled layout refused: status {})".format(st)
            return None
        G.pt = dict(struct=lay, tensors=t, nb=nb)
        G.info["spmm"] = "tiled"
        return G.pt

    def cheby_step(self, G, p, x_full, x_row_off, z, y, r, alpha, beta, gamma, coef, dots=None):
        pt = self.pt_layout(G)
        if pt is not None:
            import ctypes as C

            check(
                self.lib.meld_pt_cheby_step(C.byref(pt["struct"]), ptr(G.rowptr), ptr(G.dw_dev), G.n_rows, p, ptr(x_full), x_row_off,
                                            ptr(z), ptr(y), ptr(r), float(alpha), float(beta), float(gamma), float(coef),
                                            ptr(dots), _stream()),
                "meld_pt_cheby_step",
            )
            return
        check(
            self.lib.meld_cheby_step(ptr(G.rowptr), ptr(G.col), ptr(G.val), ptr(G.dw_dev), G.n_rows, G.nnz, p, ptr(x_full),
                                     x_row_off, ptr(z), ptr(y), ptr(r), float(alpha), float(beta), float(gamma),
                                     float(coef), ptr(dots), _stream()),
            "meld_cheby_step",
        )

    def cheby_step_wide(self, G, p, x_full, x_row_off, z, y, alpha, beta, gamma):
        """One recurrence step on a wide row-major signal [rows, p], 1 <= p <= 64 (``meld_cheby_step_wide``: lanes = columns, the
        matrix streamed once for all columns)."""
        check(
            self.lib.meld_cheby_step_wide(ptr(G.rowptr), ptr(G.col), ptr(G.val), ptr(G.dw_dev), G.n_rows, int(p), ptr(x_full), int(x_row_off),
                                          ptr(z), ptr(y), float(alpha), float(beta), float(gamma), _stream()),
            "meld_cheby_step_wide",
        )

    def cheby_run(self, G, p, t_prev2, t_prev1, r, coeffs, alpha2, beta2):
        """Steps 2 .. len(coeffs) - 1 of the Chebyshev recurrence in one call (``meld_pt_cheby_run``: single GPU, tiled layout;
        the accumulator is touched every other step).  Returns False when the graph has no tiled layout (the caller steps)."""
        pt = self.pt_layout(G)
        if pt is None or getattr(G, "comm", None) is not None or G.row_begin != 0:
            return False
        import ctypes as C

        c = np.ascontiguousarray(coeffs, dtype=np.float64)
        check(
            self.lib.meld_pt_cheby_run(C.byref(pt["struct"]), ptr(G.rowptr), ptr(G.dw_dev), G.n_rows, p, ptr(t_prev2), ptr(t_prev1), ptr(r),
                                       c.ctypes.data_as(C.c_void_p), int(c.shape[0]), float(alpha2), float(beta2), None, _stream()),
            "meld_pt_cheby_run",
        )
        return True

    @staticmethod
    def _slice_begin(G, comm):
        """First row of this rank's slice of the full-length vectors: rank * rows_pad -- also for a rank beyond the last cell
        (``G.row_begin`` is clipped to N there; it owns no rows but still takes part in every collective)."""
        begin = int(comm.rank) * int(G.rows_pad)
        assert G.n_rows == 0 or begin == G.row_begin, (begin, G.row_begin)
        return begin

    def cheby_run_sharded(self, G, p, t_prev2, t_prev1, r, coeffs, alpha2, beta2):
        """Steps 2 .. len(coeffs) - 1 on a row shard in one call (``meld_cheby_run_sharded``: the local rows' kernel and the
        all-gather of the new slice enqueued back to back from C on the library's own RCCL communicator).  Returns None when
        the graph's communicator offers no RCCL handle (gloo, host-staged test collectives: the caller steps from Python),
        else 1 / 0: whether ``t_prev1`` / ``t_prev2`` holds the last T."""
        comm = getattr(G, "comm", None)
        handle = comm.rccl() if comm is not None and hasattr(comm, "rccl") else None
        if handle is None:
            return None
        import ctypes as C

        pt = self.pt_layout(G)
        c = np.ascontiguousarray(coeffs, dtype=np.float64)
        last = C.c_int(1)
        check(
            self.lib.meld_cheby_run_sharded(handle, C.byref(pt["struct"]) if pt is not None else None, ptr(G.rowptr), ptr(G.col), ptr(G.val),
                                            ptr(G.dw_dev), G.n_rows, G.nnz, G.rows_pad, self._slice_begin(G, comm), p, ptr(t_prev2), ptr(t_prev1), ptr(r),
                                            c.ctypes.data_as(C.c_void_p), int(c.shape[0]), float(alpha2), float(beta2), C.byref(last), _stream()),
            "meld_cheby_run_sharded",
        )
        return int(last.value)

    def lanczos_steps_sharded(self, G, V, state, acc, alphas, betas, it_begin, n_iter):
        """Iterations of the one-reduction Lanczos recurrence on a row shard in one call (``meld_lanczos_steps_sharded``); False
        when the communicator offers no RCCL handle (the caller runs the phases from Python)."""
        comm = getattr(G, "comm", None)
        handle = comm.rccl() if comm is not None and hasattr(comm, "rccl") else None
        if handle is None:
            return False
        import ctypes as C

        pt = self.pt_layout(G)
        check(
            self.lib.meld_lanczos_steps_sharded(handle, C.byref(pt["struct"]) if pt is not None else None, ptr(G.rowptr), ptr(G.col), ptr(G.val),
                                                ptr(G.dw_dev), G.n_rows, G.nnz, G.rows_pad, self._slice_begin(G, comm), ptr(V[0]), ptr(V[1]), ptr(V[2]),
                                                ptr(state), ptr(acc), ptr(alphas), ptr(betas), int(it_begin), int(n_iter), _stream()),
            "meld_lanczos_steps_sharded",
        )
        return True

    def lanczos_steps(self, G, V, state, alphas, betas, it_begin, n_iter, scratch, stop=None):
        """Iterations [it_begin, it_begin + n_iter) of the device-resident Lanczos recurrence
        (``meld_lanczos_steps``): V is a [3, N] buffer of rotating vectors.  ``stop`` (int32 device tensor, optional): a launch
        that finds it nonzero does nothing (tiled layout only; the CSR kernels of small graphs run their batch out)."""
        pt = self.pt_layout(G)
        if pt is not None:
            import ctypes as C

            check(
                self.lib.meld_pt_lanczos_steps(C.byref(pt["struct"]), ptr(G.rowptr), ptr(G.dw_dev), G.n_rows, ptr(V[0]), ptr(V[1]),
                                               ptr(V[2]), ptr(state), ptr(alphas), ptr(betas), int(it_begin), int(n_iter),
                                               ptr(scratch), ptr(stop) if stop is not None else None, _stream()),
                "meld_pt_lanczos_steps",
            )
            return
        check(
            self.lib.meld_lanczos_steps(ptr(G.rowptr), ptr(G.col), ptr(G.val), ptr(G.dw_dev), G.n_rows, G.nnz, ptr(V[0]), ptr(V[1]),
                                        ptr(V[2]), ptr(state), ptr(alphas), ptr(betas), int(it_begin), int(n_iter), ptr(scratch), _stream()),
            "meld_lanczos_steps",
        )

    # the same iteration as four stream-ordered phases (row-sharded driver; see filter._lanczos_lmax_phases)
    def lanczos_spmv(self, G, x_full, z_local, y_local, state, dots):
        pt = self.pt_layout(G)
        if pt is not None:
            import ctypes as C

            check(self.lib.meld_pt_lanczos_spmv(C.byref(pt["struct"]), ptr(G.rowptr), ptr(G.dw_dev), G.n_rows, ptr(x_full), G.row_begin,
                                                ptr(z_local), ptr(y_local), ptr(state), ptr(dots), _stream()), "meld_pt_lanczos_spmv")
            return
        check(self.lib.meld_lanczos_spmv(ptr(G.rowptr), ptr(G.col), ptr(G.val), ptr(G.dw_dev), G.n_rows, G.nnz, ptr(x_full), G.row_begin,
                                         ptr(z_local), ptr(y_local), ptr(state), ptr(dots), _stream()), "meld_lanczos_spmv")

    def lanczos_alpha(self, state, dots, nrm2, alphas, it):
        check(self.lib.meld_lanczos_alpha(ptr(state), ptr(dots), ptr(nrm2), ptr(alphas), int(it), _stream()), "meld_lanczos_alpha")

    def lanczos_axpy(self, x_local, y_local, state, nrm2):
        check(self.lib.meld_lanczos_axpy(ptr(x_local), ptr(y_local), int(x_local.shape[0]), ptr(state), ptr(nrm2), _stream()), "meld_lanczos_axpy")

    def lanczos_beta(self, state, nrm2, dots, betas, it):
        check(self.lib.meld_lanczos_beta(ptr(state), ptr(nrm2), ptr(dots), ptr(betas), int(it), _stream()), "meld_lanczos_beta")

    # one-reduction form of the sharded iteration (filter._lanczos_lmax_folded)
    def lanczos_fold(self, state, acc, alphas, betas, it):
        check(self.lib.meld_lanczos_fold(ptr(state), ptr(acc), ptr(alphas), ptr(betas), int(it), _stream()), "meld_lanczos_fold")

    def lanczos_axpy3(self, y_local, u_local, u_prev_local, state, nrm2):
        check(self.lib.meld_lanczos_axpy3(ptr(y_local), ptr(u_local), ptr(u_prev_local), int(y_local.shape[0]), ptr(state), ptr(nrm2),
                                          _stream()), "meld_lanczos_axpy3")

    def scale(self, x, a, r):
        check(self.lib.meld_scale_f64(ptr(x), float(a), ptr(r), x.numel(), _stream()), "meld_scale_f64")

    def axpby(self, a, x, b, y, nrm2=None):
        check(self.lib.meld_axpby_f64(float(a), ptr(x), float(b), ptr(y), y.numel(), ptr(nrm2), _stream()), "meld_axpby_f64")


def metric_front_end(X, distance, decay):
    """The graph builders are euclidean; other metrics come in through the data.  ``distance="cosine"`` ([UPSTREAM graphtools
    ``kNNGraph(distance=...)`` -> sklearn ``NearestNeighbors(metric="cosine")``]: d(x, y) = 1 - x.y / (|x| |y|)): on unit rows
    d_cos = |x^ - y^|^2 / 2, so the neighbours are those of the euclidean search on the normalised rows, and since the kernel sees
    distances only as the ratio d / bandwidth, d_cos / bw_cos = (d_euc / bw_euc)^2:  exp(-(d_cos / bw_cos)^decay) =
    exp(-(d_euc / bw_euc)^(2 decay)) -- the euclidean graph of the normalised rows with the decay doubled, entry for entry (the
    oracle, which hands the metric to sklearn, agrees to 1e-14: tests/test_oracle.py).  Returns (X', decay', to_metric) where
    to_metric maps a euclidean bandwidth of X' to the metric's own units.
    The same argument covers "sqeuclidean" (d = d_euc^2: the decay doubled, the rows as they are) and "correlation" (the cosine
    distance of the rows with their own means removed); "l2" is euclidean."""
    if distance in ("euclidean", "l2"):
        return X, decay, None
    decay2 = None if decay is None else 2 * decay
    if distance == "sqeuclidean":
        return X, decay2, (lambda bw: bw * bw)
    if distance not in ("cosine", "correlation"):
        raise NotImplementedError(
            "distance {!r} is not implemented by the MI355X graph builder (euclidean, l2, sqeuclidean, cosine, correlation)".format(distance))
    if distance == "correlation":
        X = X - X.mean(dim=1, keepdim=True)
    nrm = torch.linalg.vector_norm(X, dim=1, keepdim=True)
    if bool((nrm == 0).any()):
        raise ValueError("{} distance is undefined for {} rows".format(distance, "all-zero" if distance == "cosine" else "constant"))
    return (X / nrm).contiguous(), decay2, (lambda bw: 0.5 * bw * bw)


def _exact_bandwidth(X, rows, knn, n_refs=None):
    """Distance to the (knn+1)-th nearest cell (self included) of the given rows over ALL references (the first
    ``n_refs`` rows of X, by default all of them), exactly:
    an fp64 GEMM-form screen keeps the 4 (knn + 1) nearest, their distances are recomputed by direct differences.
    Library path for the handful of rows the exact sweep finds with an incomplete candidate list."""
    Xr = X if n_refs is None else X[: int(n_refs)]
    n2 = (Xr * Xr).sum(1)
    out = torch.empty(rows.shape[0], dtype=torch.float64, device=X.device)
    kk = min(int(Xr.shape[0]), 4 * (knn + 1))
    for lo in range(0, rows.shape[0], 256):
        r = rows[lo : lo + 256]
        Xq = X[r]
        d2 = (Xq * Xq).sum(1)[:, None] + n2[None, :] - 2.0 * (Xq @ Xr.T)
        cand = torch.topk(d2, kk, dim=1, largest=False).indices.contiguous()
        # the candidates' distances in the SWEEP's arithmetic (meld_knn_pair_distances: the summation order of csrc/refine.hip): the sweep counts
        # the references strictly closer than the bandwidth with its own summation order, and confirms a bandwidth ranked from
        # these by construction.  (A library norm, two ulps down, was flagged again at d = 52: "could not settle".)
        dist = torch.empty(cand.shape, dtype=torch.float64, device=X.device)
        r64 = r.to(torch.int64).contiguous()
        check(get_lib().meld_knn_pair_distances(ptr(X), int(X.shape[1]), ptr(r64), ptr(cand), int(r64.shape[0]), kk, ptr(dist), _stream()),
              "meld_knn_pair_distances")
        out[lo : lo + 256] = torch.sort(dist, dim=1).values[:, min(knn, kk - 1)]
    return out.clamp_(min=float(np.finfo(np.float64).eps))


def resolve_graph_params(N, knn, thresh, ksel):
    """Parameter clipping shared by every builder ([UPSTREAM graphtools kNNGraph.__init__])."""
    if N < 3:
        raise ValueError("need at least 3 points to build a kNN graph, got {}".format(N))
    if knn > N - 2:  # graphtools clips (with a warning)
        knn = N - 2
    thresh = float(max(thresh, np.finfo(float).eps))  # graphtools' thresh floor
    if ksel is None:
        ksel = default_ksel(knn)
    if ksel < knn + 2:
        raise NotImplementedError(
            "knn={} needs a candidate list of at least knn+2 entries but the search kernel holds at most 128".format(knn)
        )
    return int(knn), thresh, int(ksel)


def symm_code(kernel_symm, theta):
    """(mode, theta) of ``meld_csr_rows_sort_merge`` for graphtools' ``kernel_symm`` / ``theta``
    [UPSTREAM ``BaseGraph._check_symmetrization``: "+" | "*" | "mnn" | None; theta in [0, 1], 1 when not given]."""
    if kernel_symm == "+":
        return (0, 0.0)
    if kernel_symm == "*":
        return (1, 0.0)
    if kernel_symm == "mnn":
        if theta is None:
            theta = 1.0
        if not isinstance(theta, (int, float)) or theta < 0 or theta > 1:
            raise ValueError("theta {} not recognized. Expected a float between 0 and 1".format(theta))
        return (2, float(theta))
    if kernel_symm is None:
        raise NotImplementedError("kernel_symm=None (a directed kernel) is not implemented: the filter needs a symmetric Laplacian")
    raise ValueError("kernel_symm '{}' not recognized. Choose from '+', '*', 'mnn', or 'none'.".format(kernel_symm))


def build_knn_graph(X, knn=5, decay=40, thresh=1e-4, anisotropy=1, ksel=None, profile=False, force_fallback=False,
                    reorder=True, bandwidth=None, bandwidth_scale=1.0, col_stats=None, knn_max=None, kernel_symm="+", theta=None, ops=None):
    """Data [N, d] -> DeviceGraph on one GPU.  Rows A2-A5 of SURVEY.md section 8(a).

    ``X`` is a CUDA fp64 tensor [N, d] (row-major).  Stages: centre + fp32 operands, MFMA
    distance GEMM with fused top-ksel, exact fp64 refinement + alpha-decay kernel, exact sweep for
    rows whose candidate list is provably incomplete, COO emit + radix sort + merge = (K + K^T)/2,
    anisotropy, degrees.
    """
    if not (isinstance(X, torch.Tensor) and X.is_cuda and X.dtype == torch.float64 and X.dim() == 2):
        raise TypeError("build_knn_graph expects a CUDA float64 tensor [N, d]")
    X = X.contiguous()
    N, d = int(X.shape[0]), int(X.shape[1])
    ops = ops if ops is not None else HipOps(X.device)
    tm = _Timer(profile)
    knn, thresh, ksel = resolve_graph_params(N, knn, thresh, ksel)
    # [UPSTREAM graphtools kNNGraph(bandwidth=, bandwidth_scale=)]: a given bandwidth (one number or one per cell) replaces the
    # distance to the knn-th neighbour; either way it is multiplied by bandwidth_scale and floored at eps
    bw_scale = float(bandwidth_scale)
    if not (bw_scale > 0 and math.isfinite(bw_scale)):
        raise ValueError("bandwidth_scale must be positive and finite, got {!r}".format(bandwidth_scale))
    symm = symm_code(kernel_symm, theta)
    if knn_max is not None:
        knn_max = int(knn_max)
        if knn_max < knn:  # [UPSTREAM kNNGraph.__init__]
            raise ValueError("`knn_max` must be greater than or equal to `knn`")
        if knn_max + 2 > ksel:  # (the candidate list has to hold them)
            ksel = min(128, ((knn_max + 2 + 31) // 32) * 32)
            if knn_max + 2 > ksel:
                raise NotImplementedError("knn_max={} needs a candidate list beyond the 128 entries the search kernel holds".format(knn_max))
    bw_fixed = None
    if bandwidth is not None:
        if callable(bandwidth):
            raise NotImplementedError("a callable bandwidth is not implemented by the MI355X graph builder")
        b = torch.as_tensor(np.asarray(bandwidth, dtype=np.float64)).to(X.device)
        if b.dim() == 0:
            b = b.expand(N)
        if tuple(b.shape) != (N,):
            raise ValueError("bandwidth must be a number or have one entry per cell ({}), got shape {}".format(N, tuple(b.shape)))
        if not bool(torch.isfinite(b).all()) or bool((b < 0).any()):
            raise ValueError("bandwidth must be finite and non-negative")
        bw_fixed = b.contiguous().clone()

    # (the frame of the search does not depend on the order of the cells: its scatter matrix and read-back go out in front of
    # the ordering, the host decomposes it while the GPU orders: frame_axes_async)
    frame_axes = ops.frame_axes_async(X, col_stats) if (bw_fixed is None and hasattr(ops, "frame_axes_async")) else None
    perm = None
    if reorder:
        from .reorder import locality_permutation

        tm.start()
        perm = locality_permutation(X)
        if perm is not None:
            X = ops.gather_rows(X, perm)
            if bw_fixed is not None:
                bw_fixed = bw_fixed.index_select(0, perm).contiguous()
        tm.stop("reorder")

    # [UPSTREAM graphtools build_kernel_to_data] caps a row at knn_max + 1 cells only where its re-search ends at
    # ``search_knn == knn_max``; the search sizes go (knn + 1) x 6, x 36, then knn_max + 1 (x 216 always exceeds the 127 this builder
    # accepts), and a row that leaves the re-search earlier holds everything inside its radius anyway.  So the result IS the hard cap
    # -- except where 36 (knn + 1) < knn_max + 1 (knn <= 2 here) and the loop never runs because no more than N // 10 rows have
    # 6 (knn + 1) cells inside their radius (or 36 (knn + 1) >= N / 2): upstream then finishes those rows with an UNCAPPED radius
    # search, and the graph is the one without knn_max.  That is decided by a count over the uncapped rows, so this corner builds
    # the uncapped kernel first.
    k1 = knn + 1
    info = None
    if knn_max is not None and decay is not None and math.isfinite(decay) and 36 * k1 < min(knn_max + 1, N):
        trial = ops.directed_kernel_coo(X, 0, N, knn, decay, thresh, ksel, tm=tm, force_fallback=force_fallback, assemble=True,
                                        bw_scale=bw_scale, bw_fixed=bw_fixed, col_stats=col_stats, knn_max=None, symm=symm, count_rows_ge=6 * k1,
                                        frame_axes=frame_axes)
        if trial[3]["rows_with_at_least"] <= N // 10 or 36 * k1 >= N / 2:
            keys, vals, bw, info = trial
            info["knn_max_uncapped_as_upstream"] = True
            knn_max = None
        del trial
    if info is None:
        keys, vals, bw, info = ops.directed_kernel_coo(X, 0, N, knn, decay, thresh, ksel, tm=tm, force_fallback=force_fallback, assemble=True,
                                                       bw_scale=bw_scale, bw_fixed=bw_fixed, col_stats=col_stats, knn_max=knn_max, symm=symm,
                                                       frame_axes=frame_axes)
    if bw_scale != 1.0:  # (the stages record the unscaled bandwidth; the graph reports the one the kernel used)
        bw = (bw * bw_scale).clamp_(min=float(np.finfo(float).eps))
    if info.get("nnz_directed", 0) == 0:
        raise ValueError("the kernel has no off-diagonal entries; cannot build a graph")
    tm.start()
    ksum = None
    if info.get("assembled") is not None:  # (the kept candidates went straight into the row buckets)
        asm = info.pop("assembled")
        rowptr, col, val = asm[:3]
        ksum = asm[3] if len(asm) > 3 else None  # (the row sums came out of the buckets with the rows)
    else:
        rowptr, col, val = ops.assemble_rows(keys, vals, 0, N, N, symm=symm)
    del keys, vals
    tm.stop("symmetrize")
    if ksum is None:
        ksum = ops.row_sums(rowptr, val, N, 1.0)
    dw = ops.anisotropy_degrees(rowptr, col, val, N, ksum, 0, anisotropy)
    tm.stop("anisotropy_degree")

    nnz = int(col.shape[0])
    info.update(N=N, d=d, knn=knn, nnz=nnz, mean_degree=nnz / N, stage_seconds=dict(tm.t), assemble=getattr(ops, "last_assemble", None))
    G = DeviceGraph(rowptr, col, val, dw, ksum=ksum, anisotropy=anisotropy, info=info)
    G.bandwidth = bw
    G.perm = perm
    G.ops = ops
    return G
