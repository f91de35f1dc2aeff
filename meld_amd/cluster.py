"""Vertex-Frequency Clustering (reference ``meld/cluster.py:13-367``) on the device.

SURVEY.md section 8f row 2.  The reference algorithm is dense -- windows are powers of the diffusion
operator (``cluster.py:158-194``), the spectrogram is a windowed graph Fourier transform over ALL
eigenvectors of the Laplacian (``:98-156``, ``:235-236``) -- so it is O(N^3) and cannot run at the hot
path's sizes.  Two methods (``method=``):

``"dense"`` (N <= ``dense.DENSE_MAX_N``; default there): the faithful reference algorithm on dense
    PyTorch-ROCm linear algebra (fp64 rocBLAS GEMMs for the window powers and U^T (W .* s), rocSOLVER
    ``eigh`` for the Fourier basis; library calls), parity-tested against the oracle's restatement.

``"filterbank"`` (default above that; BASELINE config 5): a NEW algorithm with the same role, built on the
    hot path's recurrence kernel, not a restatement of the reference.  What the reference's spectrogram
    computes is, for vertex j, scale t and frequency k, ``|sum_i U[i,k] P^t[j,i] / c_t[i]|`` normalised
    over k (the sample indicator multiplies column j as a whole and cancels in that normalisation:
    the reference's spectrogram is a structural signature of the graph, independent of the signal --
    ``tests/test_gpu_cluster.py`` checks that on the oracle).  With the window a function of the Laplacian,
    ``h_t(L) = exp(-t L / dbar)`` (a diffusion step is ``P = I - D^-1 L``; dbar = mean kernel degree;
    precedent: a pygsp Heat filter as window, ``notebooks/meld_txclustering.ipynb`` cells 31-32), the energy
    of that signature in a band b of the spectrum is a DIAGONAL entry of a function of L,

        E[t, b, j] = sum_{k in b} h_t(lambda_k)^2 U[j,k]^2 = [ (h_t^2 g_b)(L) ]_jj ,

    with g_b a partition of unity over [0, lmax] (n_bands hat functions, log-spaced), and the signature itself
    restricted to an eigenvector u_k is |u_k[j]| h_t(lambda_k).  The low end of the spectrum -- where the smooth,
    cluster-scale structure lives and where the long windows put all their weight -- is resolved explicitly:
    n_probes random +-1 vectors are low-pass filtered through the hot path's recurrence kernel, orthonormalised,
    and Rayleigh-Ritz gives approximate eigenpairs (theta_i, u_i), i < n_probes; these are the first n_probes
    COLUMNS of the spectrogram, each treated exactly like a frequency of the reference's,
    ``sum_t tanh(|u_i[j]| h_t(theta_i) / norm_t[j])``.  The rest of the spectrum enters as n_bands more columns, the
    band amplitudes ``sum_t tanh sqrt(E_rest[t,b,j]) / norm_t[j]`` of what the Ritz subspace leaves, estimated by
    Hutchinson probing of the DEFLATED operator (Hutch++; plain probing is useless for the long windows, whose
    operators have dense rows: relative error sqrt(N / n_probes) for the constant eigenvector alone) through
    per-vertex Chebyshev moments  m_k[j] = mean_r z_r[j] ((I - QQ^T) T_k(L) z'_r)[j]  accumulated over a second pass
    of the recurrence: every (t, b) is a coefficient vector applied to them, and norm_t[j]^2 = [h_t^2(L)]_jj is
    the sum of both parts.  With n_probes = N the Ritz pairs are the eigenpairs and the first N columns ARE the
    reference-style spectrogram with heat windows (tested).  Cost: 2 chebyshev_order + 1 SpMMs of n_probes
    columns, independent of n_windows x n_bands.

PCA (``meld_amd.pca``) and a seeded k-means++ / Lloyd KMeans follow (assignment + centroid update in
``csrc/kmeans.hip``).  Same constructor, methods, checks and messages as the reference class.
"""
from __future__ import annotations

from ._options import is_set, opt

import os

import numpy as np
import pandas as pd
import torch

from . import utils

__all__ = ["VertexFrequencyCluster"]


def _l2_normalize_columns(M):
    nrm = torch.linalg.vector_norm(M, dim=0, keepdim=True)
    return M / torch.where(nrm == 0, torch.ones_like(nrm), nrm)


def _lloyd_step(Y, C, lab, scratch):
    """One Lloyd iteration on the device (``meld_kmeans_assign``): labels of the nearest centroid, new centroids
    (an empty cluster keeps its centroid), inertia of the assignment -- all as device tensors, no sync."""
    from ._lib import check, get_lib, ptr

    lib = get_lib()
    n, d = Y.shape
    k = C.shape[0]
    nb = scratch["nb"]
    check(lib.meld_kmeans_assign(ptr(Y), n, d, ptr(C), k, ptr(lab), ptr(scratch["sum"]), ptr(scratch["cnt"]), ptr(scratch["in"]),
                                 nb, torch.cuda.current_stream().cuda_stream), "meld_kmeans_assign")
    sums = scratch["sum"].view(nb, k, d).sum(0)  # fixed-order reduction of the per-workgroup partials
    cnt = scratch["cnt"].view(nb, k).sum(0)
    newC = torch.where(cnt[:, None] > 0, sums / cnt.clamp(min=1.0)[:, None], C)
    return newC, scratch["in"].sum()


def _tall_gram(A, B, chunk=4096):
    """A^T B for tall-skinny A [n, r], B [n, s] (n ~ 1e6, r, s ~ 64): as a batch of chunk-row products summed at the end.
    The library GEMM runs this shape -- a 64 x 64 result with a million-long inner dimension -- on a handful of
    workgroups (54 ms at 1M x 64); split over the inner dimension it is bandwidth-bound (~1 ms)."""
    n = A.shape[0]
    nb = n // chunk
    out = torch.zeros(A.shape[1], B.shape[1], dtype=A.dtype, device=A.device)
    if nb > 0:
        out += torch.bmm(A[: nb * chunk].view(nb, chunk, -1).transpose(1, 2), B[: nb * chunk].view(nb, chunk, -1)).sum(0)
    if nb * chunk < n:
        out += A[nb * chunk :].T @ B[nb * chunk :]
    return out


def _lloyd_step_library(Y, C, lab):
    """``_lloyd_step`` for shapes beyond the HIP kernel (d > 32 or k > 64): chunked distance GEMMs, argmin (ties to the
    lowest index), index_add for the sums -- same outputs."""
    n, d = Y.shape
    k = C.shape[0]
    c2 = (C * C).sum(1)
    sums = torch.zeros(k, d, dtype=Y.dtype, device=Y.device)
    cnt = torch.zeros(k, dtype=Y.dtype, device=Y.device)
    inertia = torch.zeros((), dtype=Y.dtype, device=Y.device)
    for lo in range(0, n, 1 << 18):
        Yc = Y[lo : lo + (1 << 18)]
        d2 = (Yc * Yc).sum(1)[:, None] + c2[None, :] - 2.0 * (Yc @ C.T)
        best, idx = torch.min(d2, dim=1)
        lab[lo : lo + Yc.shape[0]] = idx.to(lab.dtype)
        sums.index_add_(0, idx, Yc)
        cnt.index_add_(0, idx, torch.ones_like(best))
        inertia = inertia + best.clamp_(min=0.0).sum()
    newC = torch.where(cnt[:, None] > 0, sums / cnt.clamp(min=1.0)[:, None], C)
    return newC, inertia


def _kmeans(Y, k, n_init=10, max_iter=300, tol=1e-4, seed=None):
    """Seeded k-means++ initialisation + Lloyd iterations on the device; best inertia of ``n_init``
    runs (sklearn's ``KMeans`` defaults; its relative tolerance is on the centre shift).  The Lloyd step is
    the HIP kernel of ``csrc/kmeans.hip`` for d <= 32, k <= 64 (what PCA(n_clusters) hands over for up to 32 clusters),
    library kernels beyond."""
    from ._lib import get_lib

    Y = Y.contiguous()
    n, d = Y.shape
    if not Y.is_cuda:
        raise RuntimeError("meld_amd needs a ROCm GPU (MI355X); there is no CPU fallback")
    if d > 32 or k > 64:
        # beyond what csrc/kmeans.hip stages in LDS (VertexFrequencyCluster(n_clusters > 32): PCA hands over n_clusters
        # features): the same Lloyd step on library kernels (rocBLAS distance GEMM + index_add), any d and k
        scratch = None
    else:
        nb = int(min(get_lib().meld_kmeans_max_blocks(), max(1, (n + 255) // 256)))
        scratch = dict(nb=nb, sum=torch.empty(nb * k * d, dtype=torch.float64, device=Y.device),
                       cnt=torch.empty(nb * k, dtype=torch.float64, device=Y.device),
                       **{"in": torch.empty(nb, dtype=torch.float64, device=Y.device)})
    lab = torch.empty(n, dtype=torch.int32, device=Y.device)
    gen = torch.Generator(device=Y.device)
    gen.manual_seed(0 if seed is None else int(seed))
    var_tol = tol * float(Y.var(dim=0, unbiased=False).mean())
    best = None
    for _ in range(max(1, int(n_init))):
        # k-means++
        first = int(torch.randint(0, n, (1,), device=Y.device, generator=gen))
        C = [Y[first]]
        d2 = ((Y - C[0]) ** 2).sum(1)
        for _c in range(1, k):
            tot = d2.sum()
            probs = d2 / tot if float(tot) > 0 else torch.full_like(d2, 1.0 / n)
            nxt = int(torch.multinomial(probs, 1, generator=gen))
            C.append(Y[nxt])
            d2 = torch.minimum(d2, ((Y - C[-1]) ** 2).sum(1))
        C = torch.stack(C).contiguous()
        for _it in range(max_iter):
            newC, _ = _lloyd_step(Y, C, lab, scratch) if scratch is not None else _lloyd_step_library(Y, C, lab)
            shift = ((newC - C) ** 2).sum()
            C = newC.contiguous()
            if (_it & 7) == 7 and float(shift) <= var_tol:  # one host sync per 8 iterations
                break
        _, inertia = _lloyd_step(Y, C, lab, scratch) if scratch is not None else _lloyd_step_library(Y, C, lab)
        inertia = float(inertia)
        if best is None or inertia < best[0]:
            best = (inertia, lab.to(torch.int64).clone())
    return best[1]


class VertexFrequencyCluster:
    """Performs Vertex Frequency clustering for data given a raw experimental signal and enhanced
    experimental signal (reference ``meld/cluster.py:13-46``; same parameters)."""

    def __init__(self, n_clusters=10, likelihood_bias=1, window_count=9, window_sizes=None, sparse=False,
                 suppress=False, random_state=None, method="auto", n_bands=16, n_probes=64, chebyshev_order=None,
                 **kwargs):
        if method not in ("auto", "dense", "filterbank"):
            raise ValueError("method value {} not recognized. Choose from ['auto', 'dense', 'filterbank']".format(method))
        # extensions of the device version (see the module docstring): which algorithm, and the resolution of the
        # filter-bank one (bands of the spectrum, random probes of the diagonal estimate, polynomial order)
        self.method = method
        self.n_bands = int(n_bands)
        self.n_probes = int(n_probes)
        self.chebyshev_order = chebyshev_order
        self.suppress = suppress
        self.sparse = sparse  # accepted for API compatibility; the device version is dense either way
        self._basewindow = None
        if window_sizes is None:
            self.window_sizes = np.power(2, np.arange(window_count))
        else:
            self.window_sizes = np.asarray(window_sizes)
        self.window_count = np.min(self.window_sizes.shape)
        self.n_clusters = n_clusters
        self.likelihood_bias = likelihood_bias
        self.random_state = random_state
        self.window = None
        self.eigenvectors = None
        self.N = None
        self.spec_hist = None
        self.spectrogram = None
        self.combined_spectrogram = None
        self.isfit = False
        self.likelihood = None
        self.sample_indicator = None
        self._sklearn_params = {"n_init": 10}
        self._sklearn_params.update(kwargs)

    # -- pieces (reference cluster.py:80-205) --------------------------------------------------------
    def _activate(self, x, alpha=1):
        return torch.tanh(alpha * torch.abs(x))

    def _compute_spectrogram(self, sample_indicator, window):
        """normalize(U^T (window .* s), axis=0)^T for one window (reference ``cluster.py:98-137``);
        ``sample_indicator``: 1-D, in the graph's internal cell order, on the device."""
        if sample_indicator.dim() != 1:
            raise ValueError("sample_indicator must be 1-dimensional. Got shape: {}".format(tuple(sample_indicator.shape)))
        C = window * sample_indicator[None, :]
        C = _l2_normalize_columns(self.eigenvectors.T @ C)
        return C.T

    def _compute_multiresolution_spectrogram(self, sample_indicator):
        spec = torch.zeros((self.windows[0].shape[1], self.eigenvectors.shape[1]), dtype=torch.float64,
                           device=self.eigenvectors.device)
        for window in self.windows:
            spec += self._activate(self._compute_spectrogram(sample_indicator, window))
        return spec

    def _compute_window(self, window, t=1):
        return _l2_normalize_columns(torch.linalg.matrix_power(window, int(t))).T

    def _compute_windows(self):
        windows = []
        cur = self._basewindow
        windows.append(_l2_normalize_columns(cur).T)
        for _ in range(len(self.window_sizes) - 1):
            cur = cur @ cur
            windows.append(_l2_normalize_columns(cur).T)
        return windows

    def _combine_spectrogram_likelihood(self, spectrogram, likelihood):
        spectrogram_n = spectrogram / np.linalg.norm(spectrogram)
        ees_n = likelihood / np.linalg.norm(likelihood, ord=2, axis=0)
        ees_n = ees_n * self.likelihood_bias
        return np.c_[spectrogram_n, ees_n]

    # -- fit (reference cluster.py:207-241) -----------------------------------------------------------
    def fit(self, G):
        """Builds the windows (powers of the diffusion operator) and the graph Fourier basis."""
        from .dense import DENSE_MAX_N

        self.graph = utils._check_pygsp_graph(G)
        G = self.graph
        self.method_ = self.method if self.method != "auto" else ("dense" if G.N <= DENSE_MAX_N else "filterbank")
        if G.n_rows != G.N and self.method_ != "filterbank":
            raise ValueError("the dense VertexFrequencyCluster needs an unsharded graph (method='filterbank' runs row-sharded)")
        if self.method_ == "filterbank":
            return self._fit_filterbank(G)
        if G.N > DENSE_MAX_N:
            raise NotImplementedError(
                "method='dense' is the reference's O(N^3) algorithm; N={} exceeds the {} cells it is offered for "
                "(use method='filterbank')".format(G.N, DENSE_MAX_N)
            )
        dev, n = G.val.device, G.N
        # dense kernel (diagonal included) and Laplacian in the graph's internal cell order
        row_of = torch.repeat_interleave(torch.arange(n, device=dev), G.rowptr[1:] - G.rowptr[:-1])
        W = torch.zeros(n, n, dtype=torch.float64, device=dev)
        W[row_of, G.col.to(torch.int64)] = G.val
        if getattr(G, "_kdiag", None) is not None:
            kdiag = G._kdiag.to(dev)
        else:
            kdiag = G.kernel_diagonal()[:n]
        K = W + torch.diag(kdiag)
        self._basewindow = K / K.sum(dim=1, keepdim=True)  # graphtools diff_op
        if np.all(np.diff(np.log2(self.window_sizes)) == 1):
            self.windows = self._compute_windows()
        else:
            self.windows = [self._compute_window(self._basewindow, t=t) for t in self.window_sizes]
        L = torch.diag(G.dw_dev[:n]) - W
        _, self.eigenvectors = torch.linalg.eigh(L)  # pygsp compute_fourier_basis
        self.N = n
        self.isfit = True
        return self

    # -- the filter-bank method (module docstring) ---------------------------------------------------------
    @staticmethod
    def _band_edges(lmax, n_bands):
        """Centres of the n_bands hat functions: 0, then log-spaced up to lmax (the low end of the spectrum, where
        the cluster structure lives, gets the resolution)."""
        eps = lmax / 64.0
        om = np.linspace(0.0, 1.0, n_bands)
        return eps * (np.power(1.0 + lmax / eps, om) - 1.0)

    @staticmethod
    def _hat(lam, centres, b):
        """b-th hat function of the partition of unity over the band centres (piecewise linear, sums to 1 on [0, lmax])."""
        c = centres[b]
        out = np.zeros_like(lam)
        if b > 0:
            lo = centres[b - 1]
            m = (lam >= lo) & (lam <= c)
            out[m] = (lam[m] - lo) / (c - lo)
        else:
            out[lam <= c] = 1.0
        if b < len(centres) - 1:
            hi = centres[b + 1]
            m = (lam > c) & (lam <= hi)
            out[m] = (hi - lam[m]) / (hi - c)
        else:
            out[lam >= c] = 1.0
        return out

    def _filterbank_functions(self, lmax, dbar):
        """[(t, b)] -> callable f(lambda) = h_t(lambda)^2 g_b(lambda), h_t = exp(-t lambda / dbar)."""
        centres = self._band_edges(lmax, self.n_bands)
        fns = []
        for t in self.window_sizes:
            for b in range(self.n_bands):
                fns.append(lambda lam, t=float(t), b=b: np.exp(-2.0 * t * lam / dbar) * self._hat(lam, centres, b))
        return fns

    def _fit_filterbank(self, G):
        """Band energies E[t, b, j] = [p_tb(L)]_jj for every window t, band b and vertex j through the hot path's
        recurrence kernel.  Plain Hutchinson probing (diag A ~ mean_r z_r .* A z_r) is hopeless for the long
        windows: h_t^2 is then a smooth low-rank operator with dense rows -- for the constant eigenvector alone the
        relative error is sqrt(N / n_probes).  So the low end of the spectrum is DEFLATED first (Hutch++):

          pass 1  Y = phi(L) Z, phi a low-pass of the longest windows' scale, on n_probes random +-1 vectors Z;
                  Q = orth(Y); Rayleigh-Ritz on Q gives approximate low eigenpairs (theta_i, u_i), whose
                  contribution  sum_i p_tb(theta_i) u_i[j]^2  is evaluated exactly;
          pass 2  the rest, (I - QQ^T) p_tb(L) (I - QQ^T) -- localised rows now --, by Hutchinson on the deflated
                  probes, through per-vertex Chebyshev moments  m_k[j] = mean_r z_r[j] ((I - QQ^T) T_k(L) z'_r)[j]:
                  every (t, b) is then a coefficient vector applied to the moments.

        Cost: 2 chebyshev_order + 1 SpMMs of n_probes columns, independent of n_windows x n_bands."""
        from .filter import _ops_of, chebyshev_coefficients

        dev, n = G.val.device, G.N
        ops = _ops_of(G)
        # Row-sharded graph (meld_amd.distributed): this rank holds the rows [r0, r0 + nl) of the padded cell range
        # [0, npad).  The tall arrays of the method (probes, their filtered images, Ritz vectors, moments) are LOCAL
        # rows; what crosses ranks is the iterate of the recurrence (an all-gather per SpMM, as in MELD's own filter)
        # and R x R matrices (Gram matrices of the CholeskyQR, Rayleigh-Ritz, the deflation: one all-reduce each).
        comm = getattr(G, "comm", None)
        npad, r0, nl, nreal = int(G.n_pad), int(G.row_begin), int(G.rows_pad), int(G.n_rows)

        def allsum(t):
            return comm.all_reduce_sum(t) if comm is not None else t

        def gram(A, B):
            return allsum(_tall_gram(A, B))

        lmax = float(G.lmax)
        kdiag = G.kernel_diagonal()
        kdiag = kdiag[r0 : r0 + nreal] if kdiag.shape[0] != nreal else kdiag
        dsum = allsum((G.dw_dev[:nreal] + kdiag).sum().reshape(1).clone())
        dbar = float(dsum) / n  # mean row sum of the kernel: P = I - D^-1 L
        self._fb = dict(lmax=lmax, dbar=dbar)
        t_max = float(np.max(self.window_sizes))
        M = self.chebyshev_order
        if M is None:  # exp(-2 t lambda / dbar) on [0, lmax] needs ~ sqrt(2 t lmax / dbar) x 4 terms; at least 64
            M = int(min(512, max(64, np.ceil(4.0 * np.sqrt(2.0 * t_max * lmax / dbar)))))
        self._fb["order"] = M
        kk = np.arange(M + 1)
        nn = M + 2
        jackson = ((nn - kk) * np.cos(np.pi * kk / nn) + np.sin(np.pi * kk / nn) / np.tan(np.pi / nn)) / nn

        def coeffs(f):
            c = chebyshev_coefficients(f, lmax, M)
            c[0] *= 0.5  # pygsp's convention: the k = 0 term enters with 1/2
            # Jackson damping: the damped expansion of a non-negative function is non-negative (no Gibbs lobes), so
            # the band energies are >= 0 by construction; the polynomial filters p_tb ARE the method's bands (they
            # resolve ~ lmax pi / M) and still sum to the damped window: sum_b p_tb = p_t.
            return c * jackson

        Cm = np.stack([coeffs(f) for f in self._filterbank_functions(lmax, dbar)])  # [T*B, M+1]
        self._fb["coeffs"] = Cm
        t_phi = max(1.0, t_max / 8.0)
        c_phi = coeffs(lambda lam: np.exp(-2.0 * t_phi * lam / dbar))
        R = int(min(self.n_probes, n))
        gen = torch.Generator(device=dev)
        gen.manual_seed(0 if self.random_state is None else int(self.random_state))
        # (every rank draws the same [n, R] signs and keeps its rows; padding rows stay zero through the recurrence)
        Zall = torch.randint(0, 2, (n, R), device=dev, generator=gen, dtype=torch.int8)
        Z = torch.zeros(nl, R, dtype=torch.float64, device=dev)
        Z[:nreal] = Zall[r0 : r0 + nreal].to(torch.float64) * 2.0 - 1.0
        del Zall
        a1 = a2 = lmax / 2.0
        from .graph import _EventSpan

        # The recurrence kernel takes the iterate two columns at a time.  Handed [n, R] row-major, every launch would
        # gather 16 bytes out of each 8 R-byte row (1M cells, R = 64: 248 us per launch instead of 110); the iterates
        # are therefore kept PAIR-MAJOR, [R / 2][npad][2], each pair a contiguous [npad, 2] array like MELD's own signal,
        # and the local rows are turned into [nl, R] once per step for the dense algebra of `visit`.
        Rp = R + (R & 1)

        def local(T):  # rows of this rank of a full-length pair-major iterate
            return T[:, r0 : r0 + nl]

        def to_pairs(A):  # local [nl, R] -> full-length [Rp / 2, npad, 2] (gathered from every rank)
            if Rp != R:
                A = torch.cat([A, torch.zeros(nl, 1, dtype=A.dtype, device=dev)], dim=1)
            P = A.view(nl, Rp // 2, 2).permute(1, 0, 2).contiguous()
            if comm is None:
                return P
            T = torch.zeros(Rp // 2, npad, 2, dtype=A.dtype, device=dev)
            for i in range(Rp // 2):
                comm.all_gather_rows(T[i], P[i])
            return T

        def from_pairs(T):  # full-length [Rp / 2, npad, 2] -> local [nl, R]
            return local(T).permute(1, 0, 2).reshape(nl, Rp)[:, :R]

        # Wide probe blocks (more than 32 columns) go through the lanes = columns kernel instead (`meld_cheby_step_wide`): the
        # iterate stays ROW-major [npad, R], the matrix is streamed once per product instead of once per column pair (1M cells,
        # 64 probes: 1.0 instead of 3.4 ms per product), one all-gather per product on a sharded graph instead of R / 2, and the
        # local rows are already in the shape the dense algebra of `visit` wants.
        wide = R > 32 and R <= 64 and hasattr(ops, "cheby_step_wide") and opt("MELD_VFC_WIDE", "1") != "0"
        self._fb["spmm"] = "wide" if wide else "pairs"
        if wide:
            def local(T):  # noqa: F811  (row-major iterate [npad, R])
                return T[r0 : r0 + nl]

            def to_pairs(A):  # noqa: F811  local [nl, R] -> full-length [npad, R]
                if comm is None:
                    return A.contiguous()
                T = torch.zeros(npad, R, dtype=A.dtype, device=dev)
                comm.all_gather_rows(T, A.contiguous())
                return T

            def from_pairs(T):  # noqa: F811
                return local(T)

        def spmm(t_in, t_zy, alpha, beta, gamma):
            if wide:
                with _EventSpan("vfc_spmm", N=n, p=R, nnz=G.nnz):
                    y_loc = t_zy[r0 : r0 + nl]
                    ops.cheby_step_wide(G, R, t_in, r0, y_loc if gamma != 0.0 else None, y_loc, alpha, beta, gamma)
                    if comm is not None:
                        comm.all_gather_rows(t_zy, y_loc)
                return
            with _EventSpan("vfc_spmm", N=n, p=R, nnz=G.nnz):
                for i in range(Rp // 2):
                    y_loc = t_zy[i, r0 : r0 + nl]
                    ops.cheby_step(G, 2, t_in[i], r0, y_loc if gamma != 0.0 else None, y_loc, None, alpha, beta, gamma, 0.0)
                    if comm is not None:
                        comm.all_gather_rows(t_zy[i], y_loc)

        def recurrence(Z0, visit):
            """visit(k, local rows of T_k(L~) Z0) for k = 0 .. M (the same fused steps as MELD's filter, R columns as R / 2 pairs).
            CONTRACT: ``visit`` must consume Tk before it returns and must not keep a reference to it -- on the wide path
            (R > 32) Tk is a live VIEW of a ping-pong buffer that the step after next overwrites (the pair path happens to hand
            out a copy); both visitors below accumulate from it at once.  The two paths add a row's products in different
            orders: features agree to rounding (1e-12), not bit for bit, across the R <= 32 / R > 32 switch
            (tests/test_gpu_cluster.py::test_filterbank_on_the_wide_kernel_equals_the_pair_path)."""
            t_old = to_pairs(Z0)
            if wide and t_old.data_ptr() == Z0.data_ptr():
                t_old = t_old.clone()  # (the recurrence writes into its buffers)
            t_cur = torch.zeros(npad, R, dtype=torch.float64, device=dev) if wide else torch.zeros(Rp // 2, npad, 2, dtype=torch.float64, device=dev)
            visit(0, Z0)
            spmm(t_old, t_cur, 1.0 / a1, -a2 / a1, 0.0)
            visit(1, from_pairs(t_cur))
            for k in range(2, M + 1):
                spmm(t_cur, t_old, 2.0 / a1, -2.0 * a2 / a1, -1.0)
                visit(k, from_pairs(t_old))
                t_old, t_cur = t_cur, t_old

        def orthonormalise(Y):
            """Q with orthonormal columns spanning those of Y [rows, R]: CholeskyQR2 -- two rounds of (Gram matrix, Cholesky
            factor, triangular solve), all tall-skinny GEMMs and R x R factorisations; on a row-sharded graph the Gram
            matrix (one all-reduce) is the only thing that crosses ranks.  The filtered probes are far from orthogonal:
            if the first factorisation breaks down (condition number beyond ~1e8) the Householder QR of the library
            takes over (one GPU only)."""
            Q = Y
            for _ in range(2):
                Gm = gram(Q, Q)
                Lc, info = torch.linalg.cholesky_ex(0.5 * (Gm + Gm.T))
                if int(info) != 0:
                    if comm is not None:
                        raise RuntimeError("CholeskyQR of the filtered probes broke down on a sharded graph; use fewer probes")
                    Qh, _ = torch.linalg.qr(Y)
                    return Qh.contiguous()
                # Q <- Q L^-T through the inverse of the small factor (a triangular solve with n right-hand sides asks the
                # library for an n-sized workspace)
                Linv = torch.linalg.solve_triangular(Lc, torch.eye(Lc.shape[0], dtype=Lc.dtype, device=Lc.device), upper=False)
                Q = (Q @ Linv.T).contiguous()
            return Q

        # pass 1: low-pass filtered probes -> basis of the low end of the spectrum -> Ritz pairs
        Y = torch.zeros_like(Z)
        recurrence(Z, lambda k, Tk: Y.add_(Tk, alpha=float(c_phi[k])))
        Q = orthonormalise(Y)
        qp = to_pairs(Q)
        lqp = torch.zeros(npad, R, dtype=torch.float64, device=dev) if wide else torch.zeros(Rp // 2, npad, 2, dtype=torch.float64, device=dev)
        spmm(qp, lqp, 1.0, 0.0, 0.0)  # L Q
        LQ = from_pairs(lqp)
        H = gram(Q, LQ)
        theta, V = torch.linalg.eigh(0.5 * (H + H.T))
        Ur = Q @ V  # approximate eigenvectors (local rows) [nl, R]
        theta = theta.clamp(0.0, lmax)
        Cd = torch.from_numpy(Cm).to(dev)
        # p_tb(theta_i) by the Clenshaw-free direct sum (R values, M + 1 terms)
        xt = (2.0 * theta / lmax - 1.0).clamp(-1.0, 1.0)
        Tm = torch.cos(torch.arange(M + 1, device=dev, dtype=torch.float64)[:, None] * torch.acos(xt)[None, :])  # [M+1, R]
        Pth = (Cd @ Tm).clamp_(min=0.0)  # [T*B, R]
        E_low = (Ur * Ur) @ Pth.T  # [nl, T*B]
        # pass 2: Hutchinson on the deflated operator
        Zd = Z - Q @ gram(Q, Z)
        mom = torch.empty(M + 1, nl, dtype=torch.float64, device=dev)

        def visit2(k, Tk):
            W = Tk - Q @ gram(Q, Tk)
            torch.mean(Z * W, dim=1, out=mom[k])

        recurrence(Zd, visit2)
        T, B = len(self.window_sizes), self.n_bands
        E_res = (Cd @ mom).T.contiguous().view(nl, T, B).clamp_(min=0.0)  # band energies outside the Ritz subspace
        E_lowb = E_low.view(nl, T, B)
        tot = (E_lowb + E_res).sum(2)  # [nl, T]: [p_t(L)]_jj, the squared norm of vertex j's window signature
        tot = torch.where(tot > 0, tot, torch.ones_like(tot))
        # columns 0 .. R-1: the Ritz vectors, each treated like a frequency of the reference spectrogram --
        # sum_t tanh(|u_i[j]| h_t(theta_i) / norm_t[j]); columns R ..: what is left per band, as amplitudes
        pt_theta = Pth.view(T, B, R).sum(1).clamp_(min=0.0)  # [T, R]: p_t(theta_i)
        ritz_feat = torch.zeros(nl, R, dtype=torch.float64, device=dev)
        absU = Ur.abs()
        for ti in range(T):
            ritz_feat += torch.tanh(absU * torch.sqrt(pt_theta[ti])[None, :] / torch.sqrt(tot[:, ti])[:, None])
        res_feat = torch.tanh(torch.sqrt(E_res / tot[:, :, None])).sum(1)  # [nl, B]
        spec = torch.cat([ritz_feat, res_feat], dim=1).contiguous()  # local rows [nl, R + B], internal cell order
        if comm is not None:  # every rank ends up with the whole spectrogram (transform / predict are host-side)
            full = torch.empty(npad, spec.shape[1], dtype=spec.dtype, device=dev)
            comm.all_gather_rows(full, spec)
            spec = full[:n]
            tot_full = torch.empty(npad, tot.shape[1], dtype=tot.dtype, device=dev)
            comm.all_gather_rows(tot_full, tot.contiguous())
            tot = tot_full[:n]
        self._fb_spectrogram = spec  # [n, R + B]
        self._fb["window_norm2"] = tot
        self._fb["ritz"] = theta
        self.eigenvectors = None
        self.N = n
        self.isfit = True
        return self

    # -- transform (reference cluster.py:243-309) -----------------------------------------------------
    def transform(self, sample_indicator, likelihood=None, center=True):
        self.sample_indicator = sample_indicator
        self.likelihood = likelihood
        if not self.isfit:
            raise ValueError("Estimator must be `fit` before running `transform`.")
        if not isinstance(self.sample_indicator, (list, tuple, np.ndarray, pd.Series, pd.DataFrame)):
            raise TypeError("`sample_indicator` must be array-like.")
        if likelihood is not None and not isinstance(self.likelihood, (list, tuple, np.ndarray, pd.Series, pd.DataFrame)):
            raise TypeError("`likelihood` must be array-like.")
        self.sample_indicator = np.array(self.sample_indicator)
        if self.N not in self.sample_indicator.shape:
            raise ValueError("At least one axis of `sample_indicator` must be" " of length `N`.")
        if likelihood is not None:
            if self.N not in self.likelihood.shape:
                raise ValueError("At least one axis of `likelihood` must be" " of length `N`.")
            if likelihood.shape != sample_indicator.shape:
                raise ValueError(
                    "`sample_indicator` and `likelihood` must have the same shape. "
                    "Got sample_indicator: {} and likelihood: {}".format(str(sample_indicator.shape), str(likelihood.shape))
                )
            self.likelihood = np.array(self.likelihood)
        if center:
            self.sample_indicator = self.sample_indicator - self.sample_indicator.mean()

        perm = getattr(self.graph, "perm", None)  # internal (locality) order -> caller's order
        if getattr(self, "method_", "dense") == "filterbank":
            # The reference's spectrogram depends on the indicator only through its zero pattern (the indicator
            # multiplies a whole column, which is then normalised): rows of zero entries are zero, every other
            # row is the structural signature; one copy per indicator column, as the reference stacks them.
            spec = self._fb_spectrogram
            if perm is not None:
                out = torch.empty_like(spec)
                out[perm] = spec
                spec = out
            spec = spec.cpu().numpy()
            ind = self.sample_indicator if self.sample_indicator.ndim > 1 else self.sample_indicator[:, None]
            if ind.shape[0] != self.N:
                ind = ind.T
            self.spectrogram = np.hstack([spec * (ind[:, i] != 0)[:, None] for i in range(ind.shape[1])])
            if self.likelihood is not None:
                self.combined_spectrogram = self._combine_spectrogram_likelihood(self.spectrogram, self.likelihood)
            return self.spectrogram
        dev = self.eigenvectors.device

        def to_internal(v):
            t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64)).to(dev)
            return t if perm is None else t.index_select(0, perm)

        def to_caller(spec):
            if perm is None:
                return spec.cpu().numpy()
            out = torch.empty_like(spec)
            out[perm] = spec
            return out.cpu().numpy()

        if self.sample_indicator.ndim == 1:
            self.spectrogram = to_caller(self._compute_multiresolution_spectrogram(to_internal(self.sample_indicator)))
        else:
            self.spectrogram = np.hstack([
                to_caller(self._compute_multiresolution_spectrogram(to_internal(self.sample_indicator[:, i])))
                for i in range(self.sample_indicator.shape[1])
            ])
        if self.likelihood is not None:
            self.combined_spectrogram = self._combine_spectrogram_likelihood(self.spectrogram, self.likelihood)
        return self.spectrogram

    def fit_transform(self, G, sample_indicator, likelihood=None, **kwargs):
        self.fit(G, **kwargs)
        return self.transform(sample_indicator, likelihood, **kwargs)

    # -- predict (reference cluster.py:315-357) --------------------------------------------------------
    def predict(self, n_clusters=None, **kwargs):
        if n_clusters is not None:
            self.n_clusters = n_clusters
        if not self.isfit:
            raise ValueError("Estimator is not fit. " "Call VertexFrequencyCluster.fit().")
        if self.spectrogram is None:
            raise ValueError("Estimator is not transformed. " "Call VertexFrequencyCluster.transform().")
        data = self.spectrogram if self.combined_spectrogram is None else self.combined_spectrogram
        from .pca import pca_project

        params = dict(self._sklearn_params)
        params.update(kwargs)
        dev = self.graph.val.device
        Y = pca_project(torch.from_numpy(np.ascontiguousarray(data, dtype=np.float64)).to(dev), self.n_clusters)
        lab = _kmeans(Y, self.n_clusters, n_init=params.get("n_init", 10), max_iter=params.get("max_iter", 300),
                      tol=params.get("tol", 1e-4), seed=params.get("random_state", self.random_state)).cpu().numpy()
        values = self.likelihood if self.likelihood is not None else self.sample_indicator
        # scprep.utils.sort_clusters_by_values: clusters relabelled by ascending mean of the values
        values = np.asarray(values, dtype=np.float64)
        if values.ndim > 1:  # (2-D indicators / likelihoods: the mean over all columns, as np.mean does in scprep)
            values = values.reshape(values.shape[0], -1).mean(1) if values.shape[0] == lab.shape[0] else values.mean(0)
        uniq, inv = np.unique(lab, return_inverse=True)
        means = np.bincount(inv, weights=values, minlength=uniq.shape[0]) / np.bincount(inv, minlength=uniq.shape[0])
        rank = np.empty(uniq.shape[0], dtype=np.int64)
        rank[np.argsort(means, kind="stable")] = np.arange(uniq.shape[0])
        self.labels_ = rank[inv]
        return self.labels_

    def fit_predict(self, G, sample_indicator, likelihood=None, **kwargs):
        self.fit_transform(G, sample_indicator, likelihood, **kwargs)
        return self.predict()

    def set_kmeans_params(self, **kwargs):
        k = kwargs.pop("n_clusters", False)
        if k:
            self.n_clusters = k
        self._sklearn_params = kwargs
